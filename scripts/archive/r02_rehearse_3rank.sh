#!/bin/bash
# bench.py --gpus 3 with gloo on one GPU: a world size that does not divide the batch
mkdir -p gpurun_out/filt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29547 \
   bench.py --gpus 3 --backend gloo --steps 9 --warmup 3 --shard-exchange both > gpurun_out/filt/three_rank_gloo.json 2> gpurun_out/filt/three_rank_gloo.err
echo "rc=$?"
grep "\[bench\]" gpurun_out/filt/three_rank_gloo.err | tail -4
python - <<'PY'
import json
for l in open("gpurun_out/filt/three_rank_gloo.json"):
    if l.startswith("{"):
        j = json.loads(l); ls = j["list_sharded"]
        print("value", round(j["value"]), j["scaling"], "rows", ls.get("identical_rows_vs_replica"), ls.get("error"),
              ls.get("steps_coalesced_per_exchange"), ls.get("code_chunks_per_rank"))
        f = ls.get("filtered_exchange")
        if f: print("filtered", round(f["queries_per_s"]), "rows", f["identical_rows_vs_replica"], f["exchange"]["bytes_ratio"])
PY
