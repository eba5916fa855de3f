# round 3 A/B on one box: two front streams (batches alternate), with 4 and 8 hardware queues
R=$PWD; O=$R/gpurun_out/r03_ab9; mkdir -p $O
run() { name=$1; shift
  env "$@" > $O/$name.json 2> $O/$name.err
  python3 - $O/$name.json $name <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = lambda d: {k: round(v, 3) for k, v in d.items()}
print(sys.argv[2].ljust(20), "ms", round(j["ms_per_step"], 3), r(j["stage_ms"]), flush=True)
PY
}
B="--shard none --traffic none --profile-only --steps 50"
run base X=1 python bench.py $B &&
run front2 TINYKNN_FRONT_STREAMS=2 python bench.py $B &&
run front2_q8 TINYKNN_FRONT_STREAMS=2 GPU_MAX_HW_QUEUES=8 python bench.py $B &&
run front1_q8 GPU_MAX_HW_QUEUES=8 python bench.py $B &&
run front2_q6 TINYKNN_FRONT_STREAMS=2 GPU_MAX_HW_QUEUES=6 python bench.py $B &&
run front2_q8_pred TINYKNN_FRONT_STREAMS=2 GPU_MAX_HW_QUEUES=8 TINYKNN_REPLAY_PRED=1 python bench.py $B
