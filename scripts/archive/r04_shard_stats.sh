# round 4: rocprofv3 kernel stats of the one-rank list-sharded leg (dense exchange, forced RCCL)
R=$PWD; O=$R/gpurun_out/r04/shard_stats; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 60 --warmup 5 --shard lists --force-collectives --shard-exchange dense --no-cpu --sweep none --traffic none --no-hbm-leg --recall-sample 10 > $O/bench.json 2> $O/bench.err
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv; rm -rf $O/kt
cd $R
python3 - $O <<'PY'
import csv, json, sys
O = sys.argv[1]
j = json.loads([l for l in open(f"{O}/bench.json") if l.startswith("{")][-1])
print("unsharded", j["value"], "sharded", j["list_sharded"]["queries_per_s"], j["list_sharded"].get("fixed_q_per_exchange", {}).get("queries_per_s"))
for r in list(csv.DictReader(open(f"{O}/kernel_stats.csv")))[:40]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {float(r['Percentage']):6.2f}")
PY
