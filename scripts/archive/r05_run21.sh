R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -q -m gpu -x --durations=5 > $O/run21_tests.txt 2>&1; echo "tests rc=$?"; tail -12 $O/run21_tests.txt
for i in 1 2; do
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --no-cpu --shard none > $O/run21_b$i.json 2> $O/run21_b$i.err
python3 - $i <<'PY'
import json, sys
i = sys.argv[1]
j = json.loads([l for l in open(f"gpurun_out/r05/run21_b{i}.json") if l.startswith("{")][-1])
print("run", i, "value", round(j["value"]), "ms", round(j["ms_per_step"], 4), "stage", {k: round(v, 3) for k, v in j["stage_ms"].items()}, "raw", round(j.get("raw_in_ids_out_queries_per_s") or 0), "graph", round((j.get("hipgraph") or {}).get("queries_per_s", 0)))
PY
done
