# round 4: build(n_probes=2) rate, pipelined mode (profile-only line of bench.py)
mkdir -p gpurun_out/r04; O=gpurun_out/r04/b2_rate.txt; : > $O
C="--steps 100 --warmup 10 --profile-only --shard none --traffic none --no-hbm-leg --no-cpu --build-probes 2"
for v in 1 2; do
  python bench.py $C 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'M q/s', 10/d['ms_per_step'], d['stage_ms'])" >> $O
done
cat $O
