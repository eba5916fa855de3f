# Measurements of record for a round: bench line, rocprofv3 kernel stats (3 / 1 batches in
# flight), HBM traffic counters (separate --pmc passes), n_probes sweep, SIFT-shaped run.
R=$PWD
O=$R/gpurun_out/final
mkdir -p $O/sweep
python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
for depth in 2 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$depth -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu --shard none --recall-sample 10 --pipeline $depth > $O/kt${depth}_bench.json 2> $O/kt$depth.err
  f=$(find $O/kt$depth -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_pipeline$depth.csv; rm -rf $O/kt$depth
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --shard none --recall-sample 10 --pipeline 1 > /dev/null 2> $O/pmc_$c.err
done
cd $R
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/final/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
    out[c] = {k: {"launches": len(v), "mean": sum(v) / len(v)} for k, v in acc.items()}
json.dump(out, open("gpurun_out/final/pmc_fetch_write.json", "w"), indent=1)
# scan kernel: FETCH_SIZE is in 32-byte units on gfx950 after the x2 correction of the guide
# (counter unit 64 B reported as 32 B: bytes = value * 64), WRITE_SIZE in 64-byte units
def pick(d, sub):
    for k, v in d.items():
        if sub in k: return v["mean"]
    return None
f = pick(out["FETCH_SIZE"], "scan_units_kernel<1, true, 3, false>")
w = pick(out["WRITE_SIZE"], "scan_units_kernel<1, true, 3, false>")
fc = pick(out["FETCH_SIZE"], "scan_units_kernel<1, true, 3, true>")
wc = pick(out["WRITE_SIZE"], "scan_units_kernel<1, true, 3, true>")
print("list scan fetch/write KiB:", f, w, " coarse scan:", fc, wc)
json.dump({"kernels": "scan_units_kernel<1,true,3,false> (list scan) + <1,true,3,true> (coarse scan): the two jobs "
                      "the pipelined mode launches as ONE scan_units2_kernel",
           "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 5 "
                      "--warmup 1 --no-cpu --shard none --pipeline 1",
           "list_scan": {"FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w},
           "coarse_scan": {"FETCH_SIZE_KiB_per_launch": fc, "WRITE_SIZE_KiB_per_launch": wc},
           "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads "
                         "(MI355X_MICROARCH.md, HBM) -> x2; WRITE_SIZE exact; both in KiB",
           "hbm_bytes_per_launch": (2 * (f + fc) + (w + wc)) * 1024},
          open("gpurun_out/final/scan_traffic.json", "w"), indent=1)
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
for np_ in 1 5 20 50; do
  python bench.py --n-probes $np_ --shard none --cpu-sample 1000 > $O/sweep/glove_np$np_.json 2>> $O/sweep.err
done
cp $O/bench.json $O/sweep/glove_np10.json
python bench.py --data sift-like --metric euclidean --d 128 --n 1000000 --n-clusters 1000 --shard none --cpu-sample 2000 > $O/sweep/sift_np10.json 2>> $O/sweep.err
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/final/sweep/*.json")) + ["gpurun_out/final/kt2_bench.json", "gpurun_out/final/kt1_bench.json"]:
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][0])
    except Exception:
        print(f, "FAILED"); continue
    print(f.split("/")[-1], "np", j["config"]["n_probes"], "recall", round(j["config"]["recall10@10"], 3), "MQPS", round(j["value"] / 1e6, 2), "ms", round(j["ms_per_step"], 3),
          "iso_ms", round(j["isolated"]["ms_per_step"], 3), "frac", round(j["roofline"]["frac"], 3), "iso_frac", round(j["isolated"]["scan_kernel_frac_of_hbm_peak"], 3),
          "cpu", j["cpu_baseline"] and round(j["cpu_baseline"]["value"]), "parity", j["parity_vs_oracle"], "host", j.get("host_boundary", {}).get("queries_per_s"))
PY
