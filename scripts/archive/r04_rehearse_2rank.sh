#!/bin/bash
# bench.py --gpus 2 as the driver launches it, but with gloo and both ranks on the one GPU of the box: the N > 1
# code path end to end (collectives staged through the host): two-phase scan + bound all-reduce, dense and
# filtered exchange (counts on the device: equal-split all-to-all of the record regions)
O=gpurun_out/r04/two_rank; mkdir -p $O
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
   bench.py --gpus 2 --backend gloo --steps 12 --warmup 3 --shard-exchange both > $O/two_rank_gloo.json 2> $O/two_rank_gloo.err
echo "rc=$?"
grep -v amdgpu.ids $O/two_rank_gloo.err | tail -4
python3 - <<'PY'
import json
for l in open("gpurun_out/r04/two_rank/two_rank_gloo.json"):
    if l.startswith("{"):
        j = json.loads(l); ls = j["list_sharded"]
        print("value", round(j["value"]), j["scaling"], "rows", ls["identical_rows_vs_replica"], ls["exchange"]["kind"][:5],
              ls["exchange"]["all_to_all_bytes_per_rank_per_step"], ls["scan"])
        f = ls.get("filtered_exchange")
        if f: print("filtered", round(f["queries_per_s"]), "rows", f["identical_rows_vs_replica"], f["exchange"])
PY
