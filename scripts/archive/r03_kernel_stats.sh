# rocprofv3 --kernel-trace --stats of the bench's timed region (profile-only), pipelined and isolated;
# usage: scripts/r03_kernel_stats.sh <out-subdir> [bench args...]
R=$PWD; O=$R/gpurun_out/$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for depth in 2 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$depth -- python3 $R/bench.py --traffic none --pipeline $depth --profile-only --steps 50 "$@" > $O/bench_pipeline$depth.json 2> $O/kt$depth.err
  f=$(find $O/kt$depth -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_pipeline$depth.csv; rm -rf $O/kt$depth
done
cd $R
python3 - "$O" <<'PY'
import csv, json, sys
O = sys.argv[1]
for d in (2, 1):
    rows = list(csv.DictReader(open(f"{O}/kernel_stats_pipeline{d}.csv")))
    j = json.loads([l for l in open(f"{O}/bench_pipeline{d}.json") if l.startswith("{")][-1])
    print("pipeline", d, "ms_per_step", round(j["ms_per_step"], 4), "stage_ms", {k: round(v, 3) for k, v in j["stage_ms"].items()})
    for r in rows[:14]:
        print("   ", r["Name"][:70].ljust(70), r["Calls"].rjust(6), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(9), "us", r["Percentage"])
PY
