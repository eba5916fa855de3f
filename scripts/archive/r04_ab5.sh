# round 4: priorities of the front stream and of the replay streams (experiment build)
mkdir -p gpurun_out/r04; O=gpurun_out/r04/ab_front_prio.txt; : > $O
C="--steps 200 --warmup 10 --profile-only --shard none --traffic none --no-hbm-leg --no-cpu"
for v in "1 0" "1 1" "1 -1" "0 1" "1 0" "1 1"; do
  set -- $v
  echo "== TINYKNN_FRONT_PRIO=$1 TINYKNN_REPLAY_PRIO=$2" >> $O
  TINYKNN_FRONT_PRIO=$1 TINYKNN_REPLAY_PRIO=$2 python bench.py $C 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])" >> $O
done
cat $O
