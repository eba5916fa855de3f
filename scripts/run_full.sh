mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py --shard lists > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
tail -2 gpurun_out/bench_full.err
python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/bench_full.json") if l.startswith("{")][0])
print(round(j["value"]/1e6,2), round(j["ms_per_step"],3), round(j["roofline"]["frac"],3), {k:round(v,3) for k,v in j["stage_ms"].items()})
print({k:round(v,3) for k,v in j["isolated"]["stage_ms"].items()}, j["parity_vs_oracle"], j["cpu_baseline"]["value"])
print(j.get("list_sharded"))
PY
