# round 4: rescoring form under the final pipeline (2 = 32-row tiles (default), 1 = 64-row tiles, 0 = a row per lane)
mkdir -p gpurun_out/r04; O=gpurun_out/r04/ab_rescore_form.txt; : > $O
C="--steps 200 --warmup 10 --profile-only --shard none --traffic none --no-hbm-leg --no-cpu"
for v in 2 1 0 2 1; do
  echo "== --rescore-form $v" >> $O
  python bench.py $C --rescore-form $v 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])" >> $O
done
cat $O
