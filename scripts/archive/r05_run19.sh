R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_full_size_gpu.py tests/test_fuzz_gpu.py tests/test_flat_top_gpu.py tests/test_plain_scan_gpu.py tests/test_reference_behaviours_gpu.py tests/test_coalesce_gpu.py tests/test_resident_build_gpu.py -q -m gpu -x > $O/run19_tests.txt 2>&1; echo "tests rc=$?"; tail -5 $O/run19_tests.txt
for ins in 1 0 1 0; do
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --no-cpu --shard none --replay-insert $ins > $O/run19_ins$ins.json 2> $O/run19_ins$ins.err
python3 - $ins <<'PY'
import json, sys
ins = sys.argv[1]
j = json.loads([l for l in open(f"gpurun_out/r05/run19_ins{ins}.json") if l.startswith("{")][-1])
rr = (j.get("roofline") or {})
print("insert", ins, "value", round(j["value"]), "ms", round(j["ms_per_step"], 4), "stage", {k: round(v, 3) for k, v in j["stage_ms"].items()}, "iso", {k: round(v, 3) for k, v in j["isolated"]["stage_ms"].items()})
print("    replay", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (rr.get("replay") or {}).items() if k in ("insert_rounds_of_the_slowest_wave", "insert_rounds_mean_per_wave", "waves", "ns_per_round_isolated", "floor_ms", "frac", "kernel_ms_isolated")})
print("    rescore", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (rr.get("rescore") or {}).items() if k in ("achieved", "peak", "frac", "frac_timed_region", "kernel_ms_isolated")}, rr.get("replay_rescore_error"))
PY
done
