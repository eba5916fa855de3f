R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_graph_capture_gpu.py tests/test_shard_gpu.py -q -m gpu > $O/run3_tests.txt 2>&1; echo "tests rc=$?"; tail -8 $O/run3_tests.txt
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --cpu-sample 500 --py-cpu-sample 20 > $O/run3_bench.json 2> $O/run3_bench.err
echo "bench rc=$?"; grep "bench\]" $O/run3_bench.err
python3 - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r05/run3_bench.json") if l.startswith("{")][-1])
ls = j.get("list_sharded", {})
print("value", round(j["value"]), "ms", round(j["ms_per_step"], 4), "sharded", round(ls.get("queries_per_s", 0)), "ratio", round(ls.get("ratio_to_unsharded_value", 0), 3),
      "rows", ls.get("identical_rows_vs_replica"), "windows", [round(x, 2) for x in ls.get("windows_ms", [])], "drift", ls.get("window_drift_last_third_over_first_third"), "rep", ls.get("windows_repeated_after_overflow"))
print("   scan", ls.get("scan", {}).get("form", "")[:40])
print("   filtered", {k: (round(v) if isinstance(v, float) else v) for k, v in ls.get("filtered_exchange", {}).items() if k != "exchange"})
print("   fixedq", ls.get("fixed_q_per_exchange"))
for k, v in ls.items():
    if k.startswith("rank_share"):
        print("  ", k, {a: b for a, b in v.items() if a not in ("exchange", "scan", "what", "code_chunks_per_rank")})
print("   raw", j.get("raw_in_ids_out_queries_per_s"), "hipgraph", {k: v for k, v in (j.get("hipgraph") or {}).items() if k in ("queries_per_s", "identical_to_stream_launch", "error")})
PY
