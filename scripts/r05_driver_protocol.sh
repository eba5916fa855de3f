# round 5: the driver's own command, timed, its line kept under gpurun_out/r05b/ (copied to profiles/r05/bench_default.json)
R=$PWD; O=$R/gpurun_out/r05b; mkdir -p $O
t0=$(date +%s)
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
echo "rc=$? seconds=$(( $(date +%s) - t0 ))"; grep "bench\]" $O/bench_default.err | tail -5
python3 - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r05b/bench_default.json") if l.startswith("{")][-1])
ls = j.get("list_sharded", {})
print("value", round(j["value"]), "ms", round(j["ms_per_step"], 4), "recall", j["config"]["recall10@10"], "parity", j["parity_vs_oracle"], "cpu", round(j["cpu_baseline"]["value"]))
r = j["roofline"]
print("roofline", r["bound"], "frac", round(r["frac"], 4), "traffic", r.get("traffic"), "hbm_frac_measured", r.get("hbm_frac_measured"))
print("  replay", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (r.get("replay") or {}).items() if k in ("frac", "floor_ms", "kernel_ms_isolated", "insert_rounds_of_the_slowest_wave", "ns_per_round_isolated")})
print("  rescore", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (r.get("rescore") or {}).items() if k in ("frac", "achieved", "peak", "kernel_ms_isolated")}, r.get("replay_rescore_error"))
print("sharded", round(ls.get("queries_per_s", 0)), "ratio", round(ls.get("ratio_to_unsharded_value", 0), 3), "rows", ls.get("identical_rows_vs_replica"), "drift", ls.get("window_drift_last_third_over_first_third"), "rep", ls.get("windows_repeated_after_overflow"), "co", ls.get("steps_coalesced_per_exchange"), "depth", ls.get("batches_in_flight"))
print("   filtered", {k: (round(v) if isinstance(v, float) else v) for k, v in ls.get("filtered_exchange", {}).items() if k != "exchange"})
print("   fixedq", ls.get("fixed_q_per_exchange"))
for k, v in ls.items():
    if k.startswith("rank_share"):
        print("  ", k, {a: b for a, b in v.items() if a not in ("exchange", "scan", "what", "code_chunks_per_rank", "windows_ms")})
print("top-level scalars", {k: v for k, v in j.items() if k.startswith(("raw_in", "list_sharded_ratio", "rank_share", "roofline_re"))})
print("sweep", {k: (round(v["queries_per_s"]) if isinstance(v, dict) and "queries_per_s" in v else v) for k, v in (j.get("sweep") or {}).items()} if isinstance(j.get("sweep"), dict) else j.get("sweep"))
print("hipgraph", {k: v for k, v in (j.get("hipgraph") or {}).items() if k in ("queries_per_s", "identical_to_stream_launch", "error")})
PY
