# round 4: more hardware queues for HIP streams (GPU_MAX_HW_QUEUES, ROCm runtime; default 4) x replay streams (--pipeline)
mkdir -p gpurun_out/r04; O=gpurun_out/r04/ab_hw_queues.txt; : > $O
C="--steps 200 --warmup 10 --profile-only --shard none --traffic none --no-hbm-leg --no-cpu"
for v in "4 2" "8 2" "8 3" "8 4" "6 3" "4 2" "8 2"; do
  set -- $v
  echo "== GPU_MAX_HW_QUEUES=$1 --pipeline $2" >> $O
  GPU_MAX_HW_QUEUES=$1 python bench.py $C --pipeline $2 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])" >> $O
done
cat $O
