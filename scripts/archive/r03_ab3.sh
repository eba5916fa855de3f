# round 3 A/B on one box (default workload, pipelined): what the front stream carries
R=$PWD; O=$R/gpurun_out/r03_ab3; mkdir -p $O
run() { # name, env..., -- bench args
  name=$1; shift
  env "$@" > $O/$name.json 2> $O/$name.err
  python3 - $O/$name.json $name <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = lambda d: {k: round(v, 2) for k, v in d.items()}
print(sys.argv[2].ljust(28), "ms", round(j["ms_per_step"], 3), "host", round(j["host_enqueue_ms_per_step"], 3), r(j["stage_ms"]), flush=True)
PY
}
B="--shard none --traffic none --profile-only --steps 50"
run base X=1 python bench.py $B &&
run tables1 TINYKNN_TABLES_STREAM=1 python bench.py $B &&
run tables2 TINYKNN_TABLES_STREAM=2 python bench.py $B &&
run tables1_lanes32 TINYKNN_TABLES_STREAM=1 TINYKNN_REPLAY_LANES_PLAIN=32 python bench.py $B &&
run depth3 X=1 python bench.py $B --pipeline 3 &&
run depth3_tables1 TINYKNN_TABLES_STREAM=1 python bench.py $B --pipeline 3
