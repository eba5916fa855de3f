# round 3 A/B on one box (default workload, pipelined): co-residency of the plain kernel with the replay waves
R=$PWD; O=$R/gpurun_out/r03_ab2; mkdir -p $O
run() { # name, env..., -- bench args
  name=$1; shift
  env "$@" > $O/$name.json 2> $O/$name.err
  python3 - $O/$name.json $name <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = lambda d: {k: round(v, 2) for k, v in d.items()}
print(sys.argv[2].ljust(28), "ms", round(j["ms_per_step"], 3), r(j["stage_ms"]), "iso", r(j["isolated_stage_ms"]), flush=True)
PY
}
B="--shard none --traffic none --profile-only --steps 50"
run base X=1 python bench.py $B &&
run lean TINYKNN_PLAIN_LEAN=1 python bench.py $B &&
run waves2 TINYKNN_REPLAY_WAVES=2 python bench.py $B &&
run waves3 TINYKNN_REPLAY_WAVES=3 python bench.py $B &&
run lanes32 TINYKNN_REPLAY_LANES_PLAIN=32 python bench.py $B &&
run lean_waves3 TINYKNN_PLAIN_LEAN=1 TINYKNN_REPLAY_WAVES=3 python bench.py $B &&
run base2 X=1 python bench.py $B
