# same-box A/B/C: previous commit's library, the current one, a third build (scripts/micro/bin/libtinyknn_hip_$THIRD.so)
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
    print(v, i, "M_qps", round(j["value"] / 1e6, 2), "ms", round(j["ms_per_step"], 4))
except Exception as e:
    print(v, i, "failed", repr(e)); print(open(f"/root/repo/gpurun_out/r05b/ab_{v}{i}.err").read()[-600:])
PY
  done
done
