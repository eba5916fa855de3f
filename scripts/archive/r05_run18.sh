R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_resident_build_gpu.py tests/test_c5_full_size_gpu.py tests/test_full_size_gpu.py tests/test_fuzz_gpu.py tests/test_plain_scan_gpu.py -q -m gpu -x > $O/run18_tests.txt 2>&1; echo "tests rc=$?"; tail -5 $O/run18_tests.txt
timeout -k 10 900 python bench.py --workload c5 --steps 20 --warmup 5 --traffic none --no-hbm-leg --rank-share 0 > $O/run18_c5.json 2> $O/run18_c5.err
echo "c5 rc=$?"; grep "bench\]" $O/run18_c5.err | tail -4
python3 - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r05/run18_c5.json") if l.startswith("{")][-1])
print("c5 value", round(j["value"]), "ms", round(j["ms_per_step"], 4), "parity", j.get("parity_vs_oracle"), "recall", j["config"].get("recall10@10"))
print("stage_ms", {k: round(v, 3) for k, v in j["stage_ms"].items()})
print("isolated", {k: round(v, 3) for k, v in j["isolated"]["stage_ms"].items()})
PY
