# round 3 A/B on one box: build_probes = 2 with the plain path pinned on x queries per replay wave x replay streams
R=$PWD; O=$R/gpurun_out/r03_ab5; mkdir -p $O
run() { name=$1; shift
  env "$@" > $O/$name.json 2> $O/$name.err
  python3 - $O/$name.json $name <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = lambda d: {k: round(v, 2) for k, v in d.items()}
print(sys.argv[2].ljust(28), "ms", round(j["ms_per_step"], 3), r(j["stage_ms"]), "iso", r(j["isolated_stage_ms"]), flush=True)
PY
}
BP2="--build-probes 2 --shard none --traffic none --profile-only --steps 50"
run bp2_exact_lanes32 X=1 python bench.py $BP2 &&
run bp2_plain_lanes32 TINYKNN_PLAIN_SCAN=2 python bench.py $BP2 &&
run bp2_plain_lanes16 TINYKNN_PLAIN_SCAN=2 TINYKNN_REPLAY_LANES=16 python bench.py $BP2 &&
run bp2_plain_lanes16_d3 TINYKNN_PLAIN_SCAN=2 TINYKNN_REPLAY_LANES=16 python bench.py $BP2 --pipeline 3 &&
run bp2_plain_lanes32_d3 TINYKNN_PLAIN_SCAN=2 python bench.py $BP2 --pipeline 3 &&
run bp2_plain_lanes16_head8 TINYKNN_PLAIN_SCAN=2 TINYKNN_REPLAY_LANES=16 TINYKNN_PLAIN_HEAD=8 python bench.py $BP2
