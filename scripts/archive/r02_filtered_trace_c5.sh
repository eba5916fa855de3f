# kernel statistics of the list-sharded leg at 100M x 128, W = 1, dense and filtered in one process
R=$PWD; O=$R/gpurun_out/filt; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t5 -- python3 $R/bench.py --workload c5 --shard lists --shard-exchange both --shard-coalesce 1 --shard-depth 2 --steps 20 --warmup 4 --no-cpu --no-hbm-leg --traffic none > $O/trace_c5.json 2> $O/trace_c5.err
f=$(find $O/t5 -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_c5_both.csv
rm -rf $O/t5
python3 - <<P
import json, csv
l = json.loads(open("$O/trace_c5.json").read().strip().splitlines()[-1])
ls = l["list_sharded"]
print("dense   ", round(ls["queries_per_s"]), ls["identical_rows_vs_replica"])
f = ls["filtered_exchange"]
print("filtered", round(f["queries_per_s"]), f["identical_rows_vs_replica"], f["exchange"]["bytes_ratio"])
for r in list(csv.DictReader(open("$O/kernel_stats_c5_both.csv"))):
    if "shard" in r["Name"] or "heap_replay_lanes" in r["Name"] or "scan_units" in r["Name"] or "copyBuffer" in r["Name"]:
        print(r["Name"][:56].ljust(56), r["Calls"].rjust(5), f"{float(r['AverageNs'])/1e3:9.1f} us {float(r['TotalDurationNs'])/1e6:9.2f} ms")
P
