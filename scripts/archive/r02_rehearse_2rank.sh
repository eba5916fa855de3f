#!/bin/bash
# bench.py --gpus 2 as the driver launches it, but with gloo and both ranks on the one GPU of the
# box: the N > 1 code path end to end (collectives staged through the host), dense and filtered
mkdir -p gpurun_out/filt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
   bench.py --gpus 2 --backend gloo --steps 12 --warmup 3 --shard-exchange both > gpurun_out/filt/two_rank_gloo.json 2> gpurun_out/filt/two_rank_gloo.err
echo "rc=$?"
grep -v amdgpu.ids gpurun_out/filt/two_rank_gloo.err | tail -4
python - <<'PY'
import json
for l in open("gpurun_out/filt/two_rank_gloo.json"):
    if l.startswith("{"):
        j = json.loads(l); ls = j["list_sharded"]
        print("value", round(j["value"]), j["scaling"], "rows", ls["identical_rows_vs_replica"], ls["exchange"]["kind"][:5],
              ls["exchange"]["all_to_all_bytes_per_rank_per_step"])
        f = ls.get("filtered_exchange")
        if f: print("filtered", round(f["queries_per_s"]), "rows", f["identical_rows_vs_replica"], f["exchange"]["bytes_ratio"],
                    f["exchange"]["record_bytes_per_rank_per_step"])
PY
