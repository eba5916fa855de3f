# round 3 A/B on one box: queries per replay wave (16 / 32 / 64) for repeating labels (build_probes = 2) and distinct ones
R=$PWD; O=$R/gpurun_out/r03_ab4; mkdir -p $O
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
B="--shard none --traffic none --profile-only --steps 50"
run bp2_lanes32 X=1 python bench.py $BP2 &&
run bp2_lanes16 TINYKNN_REPLAY_LANES=16 python bench.py $BP2 &&
run bp2_lanes16_d3 TINYKNN_REPLAY_LANES=16 python bench.py $BP2 --pipeline 3 &&
run bp2_lanes64 TINYKNN_REPLAY_LANES=64 python bench.py $BP2 &&
run base_lanes64 X=1 python bench.py $B &&
run base_lanes32 TINYKNN_REPLAY_LANES_PLAIN=32 python bench.py $B &&
run base_lanes16 TINYKNN_REPLAY_LANES_PLAIN=16 python bench.py $B
