#!/bin/bash
# bench.py --gpus 2 as the driver launches it, but with gloo and both ranks on the one GPU of the box: the N > 1 code path end to
# end (collectives staged through the host) after round 6's changes (compact final line, pruned sharded entry points,
# SimulatedPeers outside the package); $BP = build_probes
BP=${BP:-1}
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 800 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
   bench.py --gpus 2 --backend gloo --steps 12 --warmup 3 --shard-exchange both --build-probes $BP > $O/two_rank_gloo_b$BP.out 2> $O/two_rank_gloo_b$BP.err
echo "rc=$?"
grep -v amdgpu.ids $O/two_rank_gloo_b$BP.err | tail -4
python3 - $BP <<'PY'
import json, sys
lines = open(f"gpurun_out/r06/two_rank_gloo_b{sys.argv[1]}.out").read().splitlines()
print("stdout lines:", len(lines), "| last line bytes:", len(lines[-1]))
j = json.loads(lines[-1])
print({k: j.get(k) for k in ("metric", "value", "n_gpus", "scaling", "ms_per_step")})
print("config.parallelism:", j["config"].get("parallelism"))
print("list_sharded:", j.get("list_sharded"), "replica:", j.get("replica"))
d = json.loads([l for l in lines if l.startswith("# bench_detail ")][-1][len("# bench_detail "):])
ls = d["list_sharded"]
print("rows", ls.get("identical_rows_vs_replica"), ls.get("rows"), "filtered", {k: (ls.get("filtered_exchange") or {}).get(k) for k in ("queries_per_s", "identical_rows_vs_replica")}, "error", ls.get("error"))
PY
