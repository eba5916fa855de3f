R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_graph_capture_gpu.py tests/test_coalesce_gpu.py tests/test_hip_parity.py -q -m gpu -x > $O/run24_tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/run24_tests.txt
for i in 1 2; do
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --no-cpu --shard none > $O/run24_b$i.json 2> $O/run24_b$i.err
python3 - $i <<'PY'
import json, sys
i = sys.argv[1]
j = json.loads([l for l in open(f"gpurun_out/r05/run24_b{i}.json") if l.startswith("{")][-1])
g = j.get("hipgraph") or {}
print("run", i, "value", round(j["value"]), "one graph", round(g.get("queries_per_s", 0)), round(g.get("queries_per_s", 0) / j["value"], 3), "two graphs", g.get("two_graphs_at_once"), g.get("error"))
PY
done
