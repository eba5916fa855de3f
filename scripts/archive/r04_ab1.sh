# round 4 A/B: plain forms and batch size, same box
O=gpurun_out/r04/ab1; mkdir -p $O
run() { n=$1; shift; env "$@" python bench.py --no-cpu --shard none --recall-sample 10 --profile-only --traffic none --no-hbm-leg ${BARGS} > $O/$n.json 2> $O/$n.err; python3 - $O/$n.json $n <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[2], "ms_per_step", round(j.get("ms_per_step", -1), 4), {k: round(v, 3) for k, v in (j.get("stage_ms") or {}).items()} if isinstance(j.get("stage_ms"), dict) else "")
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
BARGS="--nq 10000" run wave_nq10k A=1
BARGS="--nq 10000" run old_nq10k TINYKNN_PLAIN_FORM=0
BARGS="--nq 20000" run wave_nq20k A=1
BARGS="--nq 30000" run wave_nq30k A=1
BARGS="--nq 10000" run wave_nq10k_again A=1
