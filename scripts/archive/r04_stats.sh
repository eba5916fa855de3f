# round 4: rocprofv3 kernel stats of the default bench command (pipelined, coalesce 2) + a plain-form A/B
R=$PWD; O=$R/gpurun_out/r04/stats_$1; mkdir -p $O
ARGS="--no-cpu --shard none --recall-sample 10 --profile-only --traffic none --no-hbm-leg"
python bench.py $ARGS > $O/plain3.json 2> $O/plain3.err
TINYKNN_PLAIN_FORM=0 python bench.py $ARGS > $O/plain0.json 2> $O/plain0.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 50 --warmup 5 $ARGS > $O/kt_bench.json 2> $O/kt.err
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv; rm -rf $O/kt
cd $R
python3 - $O <<'PY'
import csv, json, sys
O = sys.argv[1]
for n in ("plain3", "plain0", "kt_bench"):
    try:
        j = json.loads([l for l in open(f"{O}/{n}.json") if l.startswith("{")][-1])
        print(n, "ms_per_step", round(j["ms_per_step"], 4), {k: round(v, 3) for k, v in j["stage_ms"].items()})
    except Exception as e:
        print(n, "failed", e)
rows = list(csv.DictReader(open(f"{O}/kernel_stats.csv")))
for r in rows[:18]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {float(r['Percentage']):6.2f}")
PY
