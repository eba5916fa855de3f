#!/bin/bash
# A/B: table build of a call on the front stream (default) or on the batch's replay stream
mkdir -p gpurun_out/filt
for v in 0 2 0 2; do
  TINYKNN_TABLES_STREAM=$v python bench.py --profile-only --steps 300 --warmup 20 > gpurun_out/filt/ts_$v.json 2>/dev/null
  python -c "
import json;j=json.load(open('gpurun_out/filt/ts_$v.json'));print('TABLES_STREAM=$v', round(j['ms_per_step'],4), 'ms/step', round(1e4/j['ms_per_step']/1e3,2), 'M q/s; scan launch', round(j['stage_ms']['scan'],3))"
done
