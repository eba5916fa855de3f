# kernel statistics of the list-sharded leg at W = 1, dense and filtered exchange in one process
R=$PWD; O=$R/gpurun_out/filt; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/bench.py --shard lists --shard-exchange both --shard-depth 4 --steps 90 --warmup 9 --no-cpu --no-hbm-leg --traffic none > $O/trace.json 2> $O/trace.err
f=$(find $O/t -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_both.csv
rm -rf $O/t
python3 - <<P
import json, csv
l = json.loads(open("$O/trace.json").read().strip().splitlines()[-1])
ls = l["list_sharded"]
print("dense   ", round(ls["queries_per_s"]), ls["identical_rows_vs_replica"])
f = ls["filtered_exchange"]
print("filtered", round(f["queries_per_s"]), f["identical_rows_vs_replica"], f["exchange"]["bytes_ratio"])
for r in list(csv.DictReader(open("$O/kernel_stats_both.csv")))[:22]:
    print(r["Name"][:56].ljust(56), r["Calls"].rjust(5), f"{float(r['AverageNs'])/1e3:9.1f} us {float(r['TotalDurationNs'])/1e6:9.2f} ms")
P
