#!/bin/bash
# 100M x 128 on one GPU (W = 1): dense vs filtered exchange of the list-sharded leg — the bytes
# a rank would put on the links at the list sizes the filter was designed for (~10 000 rows)
mkdir -p gpurun_out/filt
python bench.py --workload c5 --shard lists --shard-exchange both --shard-coalesce 1 --shard-depth 2 \
   --no-cpu --no-hbm-leg --traffic none --steps 30 --warmup 4 > gpurun_out/filt/c5.json 2> gpurun_out/filt/c5.log
python - <<'P'
import json
l = json.loads(open("gpurun_out/filt/c5.json").read().strip().splitlines()[-1])
print("value", round(l["value"]))
ls = l["list_sharded"]
print("dense   ", round(ls["queries_per_s"]), ls["identical_rows_vs_replica"], ls["exchange"])
f = ls.get("filtered_exchange")
if f: print("filtered", round(f["queries_per_s"]), f["identical_rows_vs_replica"], f["exchange"])
P
