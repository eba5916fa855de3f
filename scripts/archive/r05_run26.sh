R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_resident_build_gpu.py tests/test_c5_full_size_gpu.py tests/test_flat_top_gpu.py tests/test_fuzz_gpu.py -q -m gpu -x > $O/run26_tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/run26_tests.txt
timeout -k 10 900 python bench.py --workload c5 --steps 20 --warmup 5 --traffic none --no-hbm-leg --rank-share 0 > $O/run26_c5.json 2> $O/run26_c5.err
echo "c5 rc=$?"
python3 - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r05/run26_c5.json") if l.startswith("{")][-1])
print("c5 value", round(j["value"]), "ms", round(j["ms_per_step"], 4), "parity", j.get("parity_vs_oracle"), "recall", j["config"].get("recall10@10"))
print("stage_ms", {k: round(v, 3) for k, v in j["stage_ms"].items()})
print("isolated", {k: round(v, 3) for k, v in j["isolated"]["stage_ms"].items()})
r = j["roofline"]
print("replay", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (r.get("replay") or {}).items() if k in ("frac", "floor_ms", "kernel_ms_isolated", "insert_rounds_of_the_slowest_wave", "ns_per_round_isolated", "segments_walked_per_wave")})
PY
timeout -k 10 600 python scripts/time_flat_top.py > $O/run26_flat_top.txt 2>&1; tail -5 $O/run26_flat_top.txt
