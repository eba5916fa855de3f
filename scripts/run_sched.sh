mkdir -p gpurun_out/sc
timeout 900 python -m pytest tests -x -q -m gpu -k "pipelined or larger" 2>&1 | tail -3
for depth in 2; do
  python bench.py --no-cpu --shard none --steps 30 --recall-sample 10 --pipeline $depth > gpurun_out/sc/D${depth}.json 2>gpurun_out/sc/D${depth}.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/sc/*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][0])
    except Exception as e:
        print(f, "FAILED"); continue
    print(f.split("/")[-1], round(j["value"]/1e6,2), round(j["ms_per_step"],3), round(j["roofline"]["frac"],3), {k:round(v,2) for k,v in j["stage_ms"].items()})
PY
