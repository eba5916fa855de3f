O=gpurun_out/replay_ab3; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --profile-only --steps 200 --warmup 10 --shard none > $O/$tag.json 2> $O/$tag.err; }
run w1 TINYKNN_REPLAY_WAVES=1
run w2 TINYKNN_REPLAY_WAVES=2
run w3 TINYKNN_REPLAY_WAVES=3
run w4 TINYKNN_REPLAY_WAVES=4
python - <<'PY'
import json
for t in ("w1", "w2", "w3", "w4"):
    j = json.load(open(f"gpurun_out/replay_ab3/{t}.json"))
    print(f"{t}: pipelined {j['ms_per_step']:.3f} ms/step = {1e4 / j['ms_per_step'] / 1e3:.2f} M q/s; pipelined heap {j['stage_ms']['heap']:.3f} scan {j['stage_ms']['scan']:.3f} coarse_heap {j['stage_ms']['coarse_heap']:.3f}; isolated heap {j['isolated_stage_ms']['heap']:.3f} coarse {j['isolated_stage_ms']['coarse_heap']:.3f}")
PY
