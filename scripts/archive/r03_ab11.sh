# round 3 A/B on one box: with the LDS-operand plain kernel the scan chain is at 0.30 ms — what bounds the 0.50 ms cycle?
R=$PWD; O=$R/gpurun_out/r03_ab11; mkdir -p $O
run() { name=$1; shift
  env "$@" > $O/$name.json 2> $O/$name.err
  python3 - $O/$name.json $name <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = lambda d: {k: round(v, 3) for k, v in d.items()}
print(sys.argv[2].ljust(24), "ms", round(j["ms_per_step"], 3), "host", round(j["host_enqueue_ms_per_step"], 3), r(j["stage_ms"]), flush=True)
PY
}
B="--shard none --traffic none --profile-only --steps 50"
F="TINYKNN_PLAIN_FORM=1"
run form1 $F python bench.py $B &&
run form1_front2 $F TINYKNN_FRONT_STREAMS=2 python bench.py $B &&
run form1_coarse32 $F TINYKNN_REPLAY_LANES_COARSE=32 python bench.py $B &&
run form1_coarse16 $F TINYKNN_REPLAY_LANES_COARSE=16 python bench.py $B &&
run form1_lanes32 $F TINYKNN_REPLAY_LANES_PLAIN=32 python bench.py $B &&
run form1_tables1 $F TINYKNN_TABLES_STREAM=1 python bench.py $B &&
run form1_d3 $F python bench.py $B --pipeline 3 &&
run form1_front2_q5 $F TINYKNN_FRONT_STREAMS=2 GPU_MAX_HW_QUEUES=5 python bench.py $B
