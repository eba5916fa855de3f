# round 4: build(n_probes=2): replay streams (pipeline depth) — the duplicate-test replay is the bound there
mkdir -p gpurun_out/r04; O=gpurun_out/r04/ab_b2_depth.txt; : > $O
C="--steps 100 --warmup 10 --profile-only --shard none --traffic none --no-hbm-leg --no-cpu --build-probes 2"
for v in 2 3 4 2 3; do
  echo "== --pipeline $v" >> $O
  python bench.py $C --pipeline $v 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'M q/s', 10/d['ms_per_step'])" >> $O
done
cat $O
