#!/bin/bash
# round 3 evidence at the final code: default bench line as the driver runs it (20 steps) and at 200 steps,
# rocprofv3 kernel stats of the timed region (pipelined / one batch in flight)
R=$PWD; O=$R/gpurun_out/r03_final; mkdir -p $O
( time python bench.py --steps 20 --warmup 5 > $O/bench_default_steps20.json 2> $O/bench_default_steps20.err ) 2> $O/time_steps20.txt
python bench.py --steps 200 --warmup 5 --no-cpu --no-hbm-leg --traffic none --shard none > $O/bench_default_steps200.json 2> $O/bench_default_steps200.err
python3 - <<'PY'
import json
for f in ("bench_default_steps20", "bench_default_steps200"):
    j = json.loads([l for l in open(f"gpurun_out/r03_final/{f}.json") if l.startswith("{")][-1])
    print(f, round(j["value"]), j["ms_per_step"], j["roofline"].get("frac"), j.get("parity_vs_oracle"), (j.get("list_sharded") or {}).get("queries_per_s"))
PY
cat $O/time_steps20.txt
bash scripts/r03_kernel_stats.sh r03_final/stats --shard none
