O=gpurun_out/replay_ab4; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --profile-only --steps 200 --warmup 10 --shard none > $O/$tag.json 2> $O/$tag.err; }
run pred0 TINYKNN_REPLAY_PRED=0
run pred1 TINYKNN_REPLAY_PRED=1
run pred0_b2 TINYKNN_REPLAY_PRED=0 TK_B=2
python bench.py --profile-only --steps 100 --warmup 10 --shard none --build-probes 2 > $O/pred0_b2.json 2> $O/pred0_b2.err
TINYKNN_REPLAY_PRED=1 python bench.py --profile-only --steps 100 --warmup 10 --shard none --build-probes 2 > $O/pred1_b2.json 2> $O/pred1_b2.err
python - <<'PY'
import json
for t in ("pred0", "pred1", "pred0_b2", "pred1_b2"):
    j = json.load(open(f"gpurun_out/replay_ab4/{t}.json"))
    print(f"{t}: pipelined {j['ms_per_step']:.3f} ms/step = {1e4 / j['ms_per_step'] / 1e3:.2f} M q/s; pipelined heap {j['stage_ms']['heap']:.3f} scan {j['stage_ms']['scan']:.3f} coarse_heap {j['stage_ms']['coarse_heap']:.3f}; isolated heap {j['isolated_stage_ms']['heap']:.3f} coarse {j['isolated_stage_ms']['coarse_heap']:.3f}")
PY
