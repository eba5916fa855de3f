#!/bin/bash
# bench.py --gpus 2 as the driver launches it, but with gloo and both ranks on the one GPU of the box: the N > 1 code
# path end to end (collectives staged through the host) after round 5's changes — one-phase plain scan + flag word,
# dense and filtered exchange; $BP = build_probes (2: labels repeat, the TWIN replay on the home ranks)
BP=${BP:-1}
O=gpurun_out/r05b; mkdir -p $O
timeout -k 10 800 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
   bench.py --gpus 2 --backend gloo --steps 12 --warmup 3 --shard-exchange both --build-probes $BP > $O/two_rank_gloo_b$BP.json 2> $O/two_rank_gloo_b$BP.err
echo "rc=$?"
grep -v amdgpu.ids $O/two_rank_gloo_b$BP.err | tail -4
python3 - $BP <<'PY'
import json, sys
for l in open(f"gpurun_out/r05b/two_rank_gloo_b{sys.argv[1]}.json"):
    if l.startswith("{"):
        j = json.loads(l); ls = j["list_sharded"]
        print("value", round(j["value"]), j["scaling"], "n_gpus", j["n_gpus"], "rows", ls.get("identical_rows_vs_replica"), ls.get("rows"), str(ls.get("scan"))[:200])
        f = ls.get("filtered_exchange")
        if f: print("filtered", round(f["queries_per_s"]), "rows", f["identical_rows_vs_replica"])
        print({k: ls.get(k) for k in ("error", "windows_repeated_after_overflow", "queries_per_s")})
PY
