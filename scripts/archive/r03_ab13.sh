# round 3 A/B on one box: build_probes = 2 with the LDS-operand plain kernel (small footprint beside the duplicate-test replays)
R=$PWD; O=$R/gpurun_out/r03_ab13; mkdir -p $O
run() { name=$1; shift
  env "$@" > $O/$name.json 2> $O/$name.err
  python3 - $O/$name.json $name <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = lambda d: {k: round(v, 2) for k, v in d.items()}
print(sys.argv[2].ljust(28), "ms", round(j["ms_per_step"], 3), r(j["stage_ms"]), flush=True)
PY
}
BP2="--build-probes 2 --shard none --traffic none --profile-only --steps 50"
run bp2_exact X=1 python bench.py $BP2 &&
run bp2_plain_form1 TINYKNN_PLAIN_SCAN=2 TINYKNN_PLAIN_FORM=1 python bench.py $BP2 &&
run bp2_plain_form1_head8 TINYKNN_PLAIN_SCAN=2 TINYKNN_PLAIN_FORM=1 TINYKNN_PLAIN_HEAD=8 python bench.py $BP2 &&
run bp2_plain_form1_head8_l16 TINYKNN_PLAIN_SCAN=2 TINYKNN_PLAIN_FORM=1 TINYKNN_PLAIN_HEAD=8 TINYKNN_REPLAY_LANES=16 python bench.py $BP2 &&
run bp2_plain_form1_head8_d3 TINYKNN_PLAIN_SCAN=2 TINYKNN_PLAIN_FORM=1 TINYKNN_PLAIN_HEAD=8 python bench.py $BP2 --pipeline 3
