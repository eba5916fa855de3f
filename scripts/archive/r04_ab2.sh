# round 4 A/B under coalesce 2: which stream binds?  (existing A/B switches)
O=gpurun_out/r04/ab2; mkdir -p $O
ARGS="--no-cpu --shard none --recall-sample 10 --profile-only --traffic none --no-hbm-leg"
run() { n=$1; shift; env "$@" python bench.py $ARGS > $O/$n.json 2> $O/$n.err; python3 - $O/$n.json $n <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[2], "ms_per_step", round(j.get("ms_per_step", -1), 4), {k: round(v, 3) for k, v in (j.get("stage_ms") or {}).items()})
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run base A=1
run tables_on_scan TINYKNN_TABLES_STREAM=3
run coarse32 TINYKNN_REPLAY_LANES_COARSE=32
run list32 TINYKNN_REPLAY_LANES_PLAIN=32
run coarse32_tables3 TINYKNN_REPLAY_LANES_COARSE=32 TINYKNN_TABLES_STREAM=3
run all32_tables3 TINYKNN_REPLAY_LANES_COARSE=32 TINYKNN_REPLAY_LANES_PLAIN=32 TINYKNN_TABLES_STREAM=3
run front2 TINYKNN_FRONT_STREAMS=2
run scan704 TINYKNN_SCAN_BLOCKS=704
run base_again A=1
