#!/bin/bash
# which of bench.py's extra legs (CPU baseline, HBM-scale scan, PMC traffic pass) disturbs the
# list-sharded leg that follows them?  W = 1, dense exchange, same steps each time
mkdir -p gpurun_out/filt
run() {
  tag=$1; shift
  python bench.py --shard lists --steps 30 --warmup 6 "$@" > gpurun_out/filt/probe_$tag.json 2> gpurun_out/filt/probe_$tag.log
  python - <<P
import json
l = json.loads(open("gpurun_out/filt/probe_$tag.json").read().strip().splitlines()[-1])
print("$tag", "value", round(l["value"]), "sharded", round(l["list_sharded"]["queries_per_s"]))
P
}
run none --no-cpu --no-hbm-leg --traffic none
run cpu --no-hbm-leg --traffic none
run hbm --no-cpu --traffic none
run traffic --no-cpu --no-hbm-leg
