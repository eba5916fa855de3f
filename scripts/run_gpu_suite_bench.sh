mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu -k "pipelined or hipgraph or larger" 2>&1 | tail -3
python bench.py --no-cpu --shard none > gpurun_out/bench_tok.json 2> gpurun_out/bench_tok.err
python bench.py --no-cpu --shard none --pipeline 4 > gpurun_out/bench_tok4.json 2>> gpurun_out/bench_tok.err
python bench.py --no-cpu --shard none --pipeline 2 > gpurun_out/bench_tok2.json 2>> gpurun_out/bench_tok.err
python - <<'PY'
import json
for f in ("bench_tok","bench_tok4","bench_tok2"):
    j=json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][0])
    print(f, round(j["value"]/1e6,2), round(j["ms_per_step"],3), round(j["roofline"]["frac"],3), {k:round(v,3) for k,v in j["stage_ms"].items()})
PY
