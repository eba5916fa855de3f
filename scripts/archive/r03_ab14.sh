# round 3 A/B on one box: table build off the front stream, after the needless wait on the front stream was removed
R=$PWD; O=$R/gpurun_out/r03_ab14; mkdir -p $O
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
run tables1 TINYKNN_TABLES_STREAM=1 python bench.py $B &&
run tables2 TINYKNN_TABLES_STREAM=2 python bench.py $B &&
run tables1_form1 TINYKNN_TABLES_STREAM=1 TINYKNN_PLAIN_FORM=1 python bench.py $B &&
run tables2_form1 TINYKNN_TABLES_STREAM=2 TINYKNN_PLAIN_FORM=1 python bench.py $B &&
run tables3_form1 TINYKNN_TABLES_STREAM=3 TINYKNN_PLAIN_FORM=1 python bench.py $B
