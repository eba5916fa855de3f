# round 3: n_probes sweep (BASELINE configs[3]) + SIFT-shaped run (configs[2]) + build_probes=2 line + plain scan off
R=$PWD; O=$R/gpurun_out/r03_sweep; mkdir -p $O
for np_ in 1 5 10 20 50; do
  python bench.py --n-probes $np_ --steps 50 --shard none --cpu-sample 1000 --no-hbm-leg --traffic none > $O/glove_np$np_.json 2>> $O/sweep.err
done
python bench.py --data sift-like --metric euclidean --d 128 --n 1000000 --n-clusters 1000 --steps 50 --shard none --cpu-sample 2000 --no-hbm-leg --traffic none > $O/sift_np10.json 2>> $O/sweep.err
for bp in 2 3 5; do python bench.py --build-probes $bp --steps 50 --shard none --cpu-sample 2000 --no-hbm-leg --traffic none > $O/glove_build_probes$bp.json 2>> $O/sweep.err; done
TINYKNN_PLAIN_SCAN=0 python bench.py --steps 50 --shard none --no-cpu --no-hbm-leg --traffic none > $O/glove_np10_plain_scan_off.json 2>> $O/sweep.err
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03_sweep/*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][0])
    except Exception:
        print(f, "FAILED"); continue
    ps = j["roofline"].get("plain_scan") or {}
    print(f.split("/")[-1], "np", j["config"]["n_probes"], "recall", round(j["config"]["recall10@10"], 3), "MQPS", round(j["value"] / 1e6, 2), "ms", round(j["ms_per_step"], 3),
          "raw MQPS", j.get("raw_in_ids_out") and round(j["raw_in_ids_out"]["queries_per_s"] / 1e6, 2), "iso_ms", round(j["isolated"]["ms_per_step"], 3), "frac", round(j["roofline"]["frac"], 3),
          "iso_frac", round(j["isolated"]["scan_kernel_frac_of_hbm_peak"], 3), "cpu", j["cpu_baseline"] and round(j["cpu_baseline"]["value"]), "parity", j["parity_vs_oracle"],
          "fill", ps.get("tile_fill") and round(ps["tile_fill"], 2), "flagged", ps.get("flagged_queries"), "graph", j["hipgraph"] and j["hipgraph"].get("queries_per_s") and round(j["hipgraph"]["queries_per_s"] / 1e6, 2))
PY
