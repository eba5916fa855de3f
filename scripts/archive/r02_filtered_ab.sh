#!/bin/bash
# dense vs filtered exchange of the list-sharded leg at W = 1 (one GPU): the cost of the
# bound / filter / expand kernels and of the host synchronisation, and the bytes that would travel
set -e
mkdir -p gpurun_out/filt
python bench.py --shard lists --shard-exchange both --steps 30 --warmup 6 > gpurun_out/filt/glove.json 2> gpurun_out/filt/glove.log
python - <<'P'
import json
l = json.loads(open("gpurun_out/filt/glove.json").read().strip().splitlines()[-1])
ls = l["list_sharded"]
print("dense   ", round(ls["queries_per_s"]), ls["identical_rows_vs_replica"])
f = ls["filtered_exchange"]
print("filtered", round(f["queries_per_s"]), f["identical_rows_vs_replica"], f["exchange"])
P
