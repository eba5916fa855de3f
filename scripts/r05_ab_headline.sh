# same-box A/B of the headline batch: the library of the previous commit (scripts/micro/bin/libtinyknn_hip_old.so,
# built from a git worktree) against the current one, alternating; bench.py's main line only
R=$PWD; O=$R/gpurun_out/r05b; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
FL="--steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --no-cpu --rank-share 0 --shard none --recall-sample 10 $EXTRA"
for i in 1 2; do
  for v in old new; do
    if [ $v = old ]; then export TINYKNN_HIP_LIB=$R/scripts/micro/bin/libtinyknn_hip_old.so; else unset TINYKNN_HIP_LIB; fi
    timeout -k 10 400 python3 $R/bench.py $FL > $O/ab_$v$i.json 2> $O/ab_$v$i.err
    python3 - $v $i <<'PY'
import json, sys
v, i = sys.argv[1], sys.argv[2]
try:
    j = json.loads([l for l in open(f"/root/repo/gpurun_out/r05b/ab_{v}{i}.json") if l.startswith("{")][-1])
    st = j.get("stage_ms") or {}
    print(v, i, "M_qps", round(j["value"] / 1e6, 2), "ms", round(j["ms_per_step"], 4), "parity", j.get("parity_vs_oracle"), "alone", {k: round(x, 3) for k, x in (j.get("stage_ms_isolated") or {}).items()})
except Exception as e:
    print(v, i, "failed", repr(e)); print(open(f"/root/repo/gpurun_out/r05b/ab_{v}{i}.err").read()[-600:])
PY
  done
done
