# round 3 A/B on one box: descriptors on the scan stream (TINYKNN_DESC_STREAM=1) x plain kernel form
R=$PWD; O=$R/gpurun_out/r03_ab16; mkdir -p $O
run() { name=$1; shift
  env "$@" > $O/$name.json 2> $O/$name.err
  python3 - $O/$name.json $name <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = lambda d: {k: round(v, 3) for k, v in d.items()}
print(sys.argv[2].ljust(22), "ms", round(j["ms_per_step"], 3), "host", round(j["host_enqueue_ms_per_step"], 3), r(j["stage_ms"]), flush=True)
PY
}
B="--shard none --traffic none --profile-only --steps 50"
run base X=1 python bench.py $B &&
run desc1 TINYKNN_DESC_STREAM=1 python bench.py $B &&
run desc1_form1 TINYKNN_DESC_STREAM=1 TINYKNN_PLAIN_FORM=1 python bench.py $B &&
run desc1_form1_l32 TINYKNN_DESC_STREAM=1 TINYKNN_PLAIN_FORM=1 TINYKNN_REPLAY_LANES_PLAIN=32 python bench.py $B &&
run desc1_form1_c32 TINYKNN_DESC_STREAM=1 TINYKNN_PLAIN_FORM=1 TINYKNN_REPLAY_LANES_COARSE=32 python bench.py $B &&
run desc1_form2 TINYKNN_DESC_STREAM=1 TINYKNN_PLAIN_FORM=2 python bench.py $B
