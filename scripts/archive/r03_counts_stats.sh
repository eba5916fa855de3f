#!/bin/bash
# rocprofv3 kernel stats of the c5 filtered leg, counts on the device vs on the host
R=$PWD; O=$R/gpurun_out/r03_counts; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in device host; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$c -- python3 $R/bench.py --workload c5 --shard lists --shard-exchange filtered --shard-counts $c --shard-coalesce 1 --shard-depth 2 --no-cpu --no-hbm-leg --traffic none --steps 10 --warmup 4 --windows 3 > $O/stats_c5_$c.json 2> $O/stats_c5_$c.log
  f=$(find $O/kt_$c -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_c5_$c.csv; rm -rf $O/kt_$c
done
cd $R
python3 - <<'PY'
import csv
for c in ("device", "host"):
    rows = list(csv.DictReader(open(f"gpurun_out/r03_counts/kernel_stats_c5_{c}.csv")))
    print(c)
    for r in rows[:26]:
        print("   ", r["Name"][:64].ljust(64), r["Calls"].rjust(6), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(9), "us", ("%.1f" % (float(r["TotalDurationNs"]) / 1e6)).rjust(9), "ms")
PY
