O=gpurun_out/c5_ab; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --workload c5 --profile-only --steps 60 --warmup 5 > $O/$tag.json 2> $O/$tag.err; }
run base X=1
run pred TINYKNN_REPLAY_PRED=1
run g768 TINYKNN_SCAN_BLOCKS=768
run g640 TINYKNN_SCAN_BLOCKS=640
python - <<'PY'
import json
for t in ("base", "pred", "g768", "g640"):
    j = json.load(open(f"gpurun_out/c5_ab/{t}.json"))
    print(f"{t}: {j['ms_per_step']:.3f} ms/step = {1e4 / j['ms_per_step'] / 1e3:.2f} M q/s; scan {j['stage_ms']['scan']:.3f} heap {j['stage_ms']['heap']:.3f}; iso scan {j['isolated_stage_ms']['scan']:.3f} heap {j['isolated_stage_ms']['heap']:.3f} coarse_heap {j['isolated_stage_ms']['coarse_heap']:.3f}")
PY
