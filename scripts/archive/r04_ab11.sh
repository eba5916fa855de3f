# round 4: wave priority (s_setprio) of the rescoring kernels, table build at 3 (experiment build)
mkdir -p gpurun_out/r04; O=gpurun_out/r04/ab_rescore_prio.txt; : > $O
C="--steps 200 --warmup 10 --profile-only --shard none --traffic none --no-hbm-leg --no-cpu"
for v in "0 0" "3 0" "3 3" "0 3" "0 0" "3 0" "3 3"; do
  set -- $v
  echo "== coarse $1 final $2" >> $O
  TINYKNN_RESCORE_PRIO_COARSE=$1 TINYKNN_RESCORE_PRIO_FINAL=$2 python bench.py $C 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])" >> $O
done
cat $O
