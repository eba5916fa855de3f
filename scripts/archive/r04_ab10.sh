# round 4: make_slots as the epilogue of the coarse rescoring (default) against the separate kernel (experiment switch)
mkdir -p gpurun_out/r04; O=gpurun_out/r04/ab_fused_slots.txt; : > $O
C="--steps 200 --warmup 10 --profile-only --shard none --traffic none --no-hbm-leg --no-cpu"
for v in sep fused sep fused sep fused; do
  echo "== $v" >> $O
  if [ $v = sep ]; then export TINYKNN_NO_FUSED_SLOTS=1; else unset TINYKNN_NO_FUSED_SLOTS; fi
  python bench.py $C 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])" >> $O
done
cat $O
