# round 3 A/B on one box: replay fetching only the blocks whose minimum passes (fewer line visits beside the scans)
R=$PWD; O=$R/gpurun_out/r03_ab8; mkdir -p $O
run() { name=$1; shift
  env "$@" > $O/$name.json 2> $O/$name.err
  python3 - $O/$name.json $name <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = lambda d: {k: round(v, 3) for k, v in d.items()}
print(sys.argv[2].ljust(20), "ms", round(j["ms_per_step"], 3), r(j["stage_ms"]), "iso heap", round(j["isolated_stage_ms"]["heap"], 3), flush=True)
PY
}
B="--shard none --traffic none --profile-only --steps 50"
run base X=1 python bench.py $B &&
run pred TINYKNN_REPLAY_PRED=1 python bench.py $B &&
run base2 X=1 python bench.py $B &&
run pred2 TINYKNN_REPLAY_PRED=1 python bench.py $B
