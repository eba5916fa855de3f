R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_fuzz_gpu.py tests/test_flat_top_gpu.py -q -m gpu -x > $O/run20_tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/run20_tests.txt
for ins in 1 0 1 0; do
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --no-cpu --shard none --replay-insert $ins > $O/run20_ins$ins.json 2> $O/run20_ins$ins.err
python3 - $ins <<'PY'
import json, sys
ins = sys.argv[1]
j = json.loads([l for l in open(f"gpurun_out/r05/run20_ins{ins}.json") if l.startswith("{")][-1])
rr = (j.get("roofline") or {})
print("insert", ins, "value", round(j["value"]), "ms", round(j["ms_per_step"], 4), "stage heap", round(j["stage_ms"]["heap"], 3), "coarse_heap", round(j["stage_ms"]["coarse_heap"], 3), "iso heap", round(j["isolated"]["stage_ms"]["heap"], 3), "iso coarse_heap", round(j["isolated"]["stage_ms"]["coarse_heap"], 3), "ns/round", round((rr.get("replay") or {}).get("ns_per_round_isolated", 0)))
PY
done
