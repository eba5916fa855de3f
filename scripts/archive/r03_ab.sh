# round 3 A/B on one box: sift-like with the automatic plain mode vs off; build_probes=2 x plain x depth
R=$PWD; O=$R/gpurun_out/r03_ab; mkdir -p $O
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
SIFT="--data sift-like --metric euclidean --d 128 --n 1000000 --n-clusters 1000 --shard none --traffic none --profile-only --steps 50"
BP2="--build-probes 2 --shard none --traffic none --profile-only --steps 50"
run sift_auto X=1 python bench.py $SIFT &&
run sift_off TINYKNN_PLAIN_SCAN=0 python bench.py $SIFT &&
run bp2_auto_d2 X=1 python bench.py $BP2 &&
run bp2_off_d2 TINYKNN_PLAIN_SCAN=0 python bench.py $BP2 &&
run bp2_auto_d3 X=1 python bench.py $BP2 --pipeline 3 &&
run bp2_off_d3 TINYKNN_PLAIN_SCAN=0 python bench.py $BP2 --pipeline 3 &&
run bp2_auto_d4 X=1 python bench.py $BP2 --pipeline 4 &&
bash scripts/r03_kernel_stats.sh r03_ab/stats_bp2 --build-probes 2 --shard none
