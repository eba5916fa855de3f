# same-box A/B(/C) of bench.py's main line: the library of an earlier commit (built from a git worktree into
# scripts/micro/bin/libtinyknn_hip_old.so, chosen through TINYKNN_HIP_LIB) against the current one (and a third build
# scripts/micro/bin/libtinyknn_hip_$THIRD.so); prints the rate and the lane replay's time alone
R=$PWD; O=$R/gpurun_out/r05b; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
FL="--steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --no-cpu --rank-share 0 --shard none --recall-sample 10 $EXTRA"
for i in 1 2; do
  for v in old new $THIRD; do
    if [ $v = new ]; then unset TINYKNN_HIP_LIB; else export TINYKNN_HIP_LIB=$R/scripts/micro/bin/libtinyknn_hip_$v.so; fi
    timeout -k 10 400 python3 $R/bench.py $FL > $O/ab_$v$i.json 2> $O/ab_$v$i.err
    python3 - $v $i <<'PY'
import json, sys
v, i = sys.argv[1], sys.argv[2]
try:
    j = json.loads([l for l in open(f"/root/repo/gpurun_out/r05b/ab_{v}{i}.json") if l.startswith("{")][-1])
    rp = (j.get("roofline") or {}).get("replay") or {}
    print(v, i, "M_qps", round(j["value"] / 1e6, 2), "ms", round(j["ms_per_step"], 4), "replay alone ms", rp.get("kernel_ms_isolated"), "rounds", rp.get("insert_rounds_of_the_slowest_wave"))
except Exception as e:
    print(v, i, "failed", repr(e)); print(open(f"/root/repo/gpurun_out/r05b/ab_{v}{i}.err").read()[-600:])
PY
  done
done
