#!/bin/bash
# bench.py --gpus 2 --build-probes 2 (the reference's default build: repeating labels, duplicate
# test in the replay) with gloo on one GPU, both exchanges
mkdir -p gpurun_out/filt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 \
   bench.py --gpus 2 --backend gloo --build-probes 2 --steps 6 --warmup 3 --shard-exchange both > gpurun_out/filt/two_rank_gloo_b2.json 2> gpurun_out/filt/two_rank_gloo_b2.err
echo "rc=$?"
grep "\[bench\]" gpurun_out/filt/two_rank_gloo_b2.err | tail -3
python - <<'PY'
import json
for l in open("gpurun_out/filt/two_rank_gloo_b2.json"):
    if l.startswith("{"):
        j = json.loads(l); ls = j["list_sharded"]
        print("value", round(j["value"]), j["scaling"], "rows", ls.get("identical_rows_vs_replica"), ls.get("error"))
        f = ls.get("filtered_exchange")
        if f: print("filtered", round(f["queries_per_s"]), "rows", f["identical_rows_vs_replica"], f["exchange"]["bytes_ratio"])
PY
