# round 5, GPU run 1: new GPU tests, then the sharded leg (one-phase scan) with rank share, then the old two-phase form (A/B),
# then a kernel trace of the sharded leg
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_graph_capture_gpu.py tests/test_shard_gpu.py tests/test_coalesce_gpu.py -x -q -m gpu > $O/run1_tests.txt 2>&1
echo "tests rc=$?" ; tail -5 $O/run1_tests.txt
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --cpu-sample 500 --py-cpu-sample 20 > $O/run1_bench_one_phase.json 2> $O/run1_bench_one_phase.err
echo "bench one-phase rc=$?"; tail -3 $O/run1_bench_one_phase.err
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --no-cpu --shard-plain 2 --rank-share 0 > $O/run1_bench_two_phase.json 2> $O/run1_bench_two_phase.err
echo "bench two-phase rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt1 -- python3 $R/bench.py --steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --no-cpu --rank-share 0 --shard-exchange dense > $O/run1_trace_bench.json 2> $O/run1_trace_bench.err
cd $R
python3 scripts/r05_shard_trace.py $O/kt1 $O/run1_shard_trace.txt | head -50
rm -rf $O/kt1
python3 - <<'PY'
import json
for f in ("run1_bench_one_phase", "run1_bench_two_phase"):
    try:
        j = json.loads([l for l in open(f"gpurun_out/r05/{f}.json") if l.startswith("{")][-1])
        ls = j.get("list_sharded", {})
        print(f, "value", round(j["value"]), "ms", round(j["ms_per_step"], 4), "sharded", round(ls.get("queries_per_s", 0)), "ratio", round(ls.get("ratio_to_unsharded_value", 0), 3),
              "rows", ls.get("identical_rows_vs_replica"), "windows", [round(x, 2) for x in ls.get("windows_ms", [])])
        print("   filtered", {k: (round(v) if isinstance(v, float) else v) for k, v in ls.get("filtered_exchange", {}).items() if k != "exchange"})
        print("   fixedq", ls.get("fixed_q_per_exchange"))
        for k, v in ls.items():
            if k.startswith("rank_share"):
                print("  ", k, {a: b for a, b in v.items() if a not in ("exchange", "scan", "what", "code_chunks_per_rank")})
        print("   parity", j.get("parity_vs_oracle"), "hipgraph", {k: v for k, v in (j.get("hipgraph") or {}).items() if k in ("queries_per_s", "identical_to_stream_launch", "error")})
    except Exception as e:
        print(f, "failed", repr(e))
PY
