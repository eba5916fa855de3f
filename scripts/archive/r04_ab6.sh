# round 4: pipeline depth (replay streams) after zero-copy pairs + high-priority front stream
mkdir -p gpurun_out/r04; O=gpurun_out/r04/ab_depth.txt; : > $O
C="--steps 200 --warmup 10 --profile-only --shard none --traffic none --no-hbm-leg --no-cpu"
for v in 2 3 4 2 3; do
  echo "== --pipeline $v" >> $O
  python bench.py $C --pipeline $v 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])" >> $O
done
cat $O
