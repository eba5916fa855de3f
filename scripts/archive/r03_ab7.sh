# round 3 A/B on one box, on top of staged rescoring: knobs that did nothing while rescoring stretched the scans
R=$PWD; O=$R/gpurun_out/r03_ab7; mkdir -p $O
run() { name=$1; shift
  env "$@" > $O/$name.json 2> $O/$name.err
  python3 - $O/$name.json $name <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = lambda d: {k: round(v, 3) for k, v in d.items()}
print(sys.argv[2].ljust(20), "ms", round(j["ms_per_step"], 3), r(j["stage_ms"]), flush=True)
PY
}
B="--shard none --traffic none --profile-only --steps 50"
run base X=1 python bench.py $B &&
run lanes32 TINYKNN_REPLAY_LANES_PLAIN=32 python bench.py $B &&
run plain_blocks384 TINYKNN_PLAIN_BLOCKS=384 python bench.py $B &&
run plain_blocks768 TINYKNN_PLAIN_BLOCKS=768 python bench.py $B &&
run scan_blocks384 TINYKNN_SCAN_BLOCKS=384 python bench.py $B &&
run scan_blocks768 TINYKNN_SCAN_BLOCKS=768 python bench.py $B &&
run waves2 TINYKNN_REPLAY_WAVES=2 python bench.py $B &&
run tables1 TINYKNN_TABLES_STREAM=1 python bench.py $B &&
run prio0 TINYKNN_REPLAY_PRIO=0 python bench.py $B &&
run base2 X=1 python bench.py $B
