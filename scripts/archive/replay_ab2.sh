O=gpurun_out/replay_ab2; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --profile-only --steps 200 --warmup 10 --shard none > $O/$tag.json 2> $O/$tag.err; }
run base TINYKNN_MERGE_EVENTS=0
run merge TINYKNN_MERGE_EVENTS=1
run merge_p2 TINYKNN_REPLAY_PRIO=2
run merge_p1 TINYKNN_REPLAY_PRIO=1
run merge_p0 TINYKNN_REPLAY_PRIO=0
python - <<'PY'
import json
for t in ("base", "merge", "merge_p2", "merge_p1", "merge_p0"):
    j = json.load(open(f"gpurun_out/replay_ab2/{t}.json"))
    print(f"{t}: pipelined {j['ms_per_step']:.3f} ms/step = {1e4 / j['ms_per_step'] / 1e3:.2f} M q/s; pipelined heap {j['stage_ms']['heap']:.3f} scan {j['stage_ms']['scan']:.3f} coarse_heap {j['stage_ms']['coarse_heap']:.3f}")
PY
