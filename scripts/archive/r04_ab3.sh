# round 4 A/B: the reference's default build (every point in two lists): plain path on / off with the wave-per-unit kernel
O=gpurun_out/r04/ab3; mkdir -p $O
ARGS="--no-cpu --shard none --recall-sample 10 --profile-only --traffic none --no-hbm-leg --build-probes 2"
run() { n=$1; shift; env "$@" python bench.py $ARGS > $O/$n.json 2> $O/$n.err; python3 - $O/$n.json $n <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[2], "ms_per_step", round(j.get("ms_per_step", -1), 4), {k: round(v, 3) for k, v in (j.get("stage_ms") or {}).items()})
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run b2_default A=1
run b2_plain_always TINYKNN_PLAIN_SCAN=2
