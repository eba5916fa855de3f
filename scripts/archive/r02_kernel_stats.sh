# rocprofv3 --kernel-trace --stats of the default bench command (the PMC child runs of the traffic leg are
# left out: a profiler inside a profiled process), summary copied to profiles/
R=$PWD; O=$R/gpurun_out/r02_stats; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for depth in 2 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$depth -- python3 $R/bench.py --traffic none --pipeline $depth > $O/bench_pipeline$depth.json 2> $O/kt$depth.err
  f=$(find $O/kt$depth -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_pipeline$depth.csv; rm -rf $O/kt$depth
done
cd $R
python3 - <<'PY'
import csv, json
for d in (2, 1):
    rows = list(csv.DictReader(open(f"gpurun_out/r02_stats/kernel_stats_pipeline{d}.csv")))
    j = json.loads([l for l in open(f"gpurun_out/r02_stats/bench_pipeline{d}.json") if l.startswith("{")][-1])
    print("pipeline", d, "value", round(j["value"] / 1e6, 2), "M q/s; bench kernel_ms", round(j["roofline"]["kernel_ms"], 4))
    for r in rows[:8]:
        print("   ", r["Name"][:60].ljust(60), r["Calls"].rjust(6), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(9), "us", r["Percentage"])
PY
