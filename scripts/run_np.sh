mkdir -p gpurun_out/np
python bench.py --no-cpu --shard none --recall-sample 10 --steps 5 > /dev/null 2>&1
for np_ in 1 5 20; do
  python bench.py --n-probes $np_ --shard none --no-cpu --recall-sample 10 > gpurun_out/np/np$np_.json 2>/dev/null
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/np/*.json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][0])
    print(f.split("/")[-1], "MQPS", round(j["value"] / 1e6, 2), "ms", round(j["ms_per_step"], 3), "frac", round(j["roofline"]["frac"], 3), {k: round(v, 2) for k, v in j["stage_ms"].items()})
PY
