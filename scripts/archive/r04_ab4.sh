# round 4: would batches of 40 000 queries (4 calls as one) beat pairs?  Same box, per-query rate.
mkdir -p gpurun_out/r04; O=gpurun_out/r04/ab_batch_size.txt; : > $O
C="--steps 200 --warmup 10 --profile-only --shard none --traffic none --no-hbm-leg --no-cpu"
for cfg in "10000 2" "20000 2" "40000 1" "20000 1" "10000 2"; do
  set -- $cfg
  echo "== nq $1 coalesce $2" >> $O
  python bench.py $C --nq $1 --coalesce $2 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'M queries/s', $1/d['ms_per_step']/1e3)" >> $O
done
cat $O
