# isolated heap stage + pipelined step, distinct (build_probes 1) and repeating labels (2)
O=gpurun_out/replay_ab; mkdir -p $O
for b in 1 2; do
  python bench.py --profile-only --steps 100 --warmup 10 --shard none --build-probes $b > $O/b$b.json 2> $O/b$b.err
done
python - <<'PY'
import json
for b in (1, 2):
    j = json.load(open(f"gpurun_out/replay_ab/b{b}.json"))
    print(f"build_probes={b}: pipelined {j['ms_per_step']:.3f} ms/step = {1e4 / j['ms_per_step'] / 1e3:.2f} M q/s; isolated heap {j['isolated_stage_ms']['heap']:.3f} coarse_heap {j['isolated_stage_ms']['coarse_heap']:.3f} scan {j['isolated_stage_ms']['scan']:.3f}; pipelined heap {j['stage_ms']['heap']:.3f} scan {j['stage_ms']['scan']:.3f}")
PY
