# round 3 A/B on one box: plain kernel with the table operand from LDS (TINYKNN_PLAIN_FORM=1: 128 registers, 31 KB, four waves per SIMD)
R=$PWD; O=$R/gpurun_out/r03_ab10; mkdir -p $O
run() { name=$1; shift
  env "$@" > $O/$name.json 2> $O/$name.err
  python3 - $O/$name.json $name <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = lambda d: {k: round(v, 3) for k, v in d.items()}
print(sys.argv[2].ljust(20), "ms", round(j["ms_per_step"], 3), r(j["stage_ms"]), "iso scan", round(j["isolated_stage_ms"]["scan"], 3), flush=True)
PY
}
B="--shard none --traffic none --profile-only --steps 50"
run form0 X=1 python bench.py $B &&
run form1 TINYKNN_PLAIN_FORM=1 python bench.py $B &&
run form2 TINYKNN_PLAIN_FORM=2 python bench.py $B &&
run form1_b384 TINYKNN_PLAIN_FORM=1 TINYKNN_PLAIN_BLOCKS=384 python bench.py $B &&
run form1_b256 TINYKNN_PLAIN_FORM=1 TINYKNN_PLAIN_BLOCKS=256 python bench.py $B &&
run form0_again X=1 python bench.py $B
