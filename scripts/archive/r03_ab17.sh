# round 3 A/B on one box: every relief at once (descriptors on the scan stream, LDS-operand plain kernel, shorter replays, a third replay stream)
R=$PWD; O=$R/gpurun_out/r03_ab17; mkdir -p $O
run() { name=$1; shift
  env "$@" > $O/$name.json 2> $O/$name.err
  python3 - $O/$name.json $name <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = lambda d: {k: round(v, 3) for k, v in d.items()}
print(sys.argv[2].ljust(26), "ms", round(j["ms_per_step"], 3), "host", round(j["host_enqueue_ms_per_step"], 3), r(j["stage_ms"]), flush=True)
PY
}
B="--shard none --traffic none --profile-only --steps 50"
E="TINYKNN_DESC_STREAM=1 TINYKNN_PLAIN_FORM=1"
run base X=1 python bench.py $B &&
run all_d3 $E python bench.py $B --pipeline 3 &&
run all_d3_l32 $E TINYKNN_REPLAY_LANES_PLAIN=32 python bench.py $B --pipeline 3 &&
run all_l32_pred $E TINYKNN_REPLAY_LANES_PLAIN=32 TINYKNN_REPLAY_PRED=1 python bench.py $B &&
run all_l32_staged0 $E TINYKNN_REPLAY_LANES_PLAIN=32 TINYKNN_RESCORE_STAGED=0 python bench.py $B
