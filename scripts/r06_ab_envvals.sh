# headline batch by the values $VALS of an environment switch $VAR of the library, taken in turn twice, same box
O=gpurun_out/r06; mkdir -p $O
FLAGS="--steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --shard none --cpu-sample 10000 $EXTRA"
for rep in 1 2; do for v in $VALS; do
  env $VAR=$v timeout -k 10 300 python bench.py $FLAGS > $O/ab_envv_$v.out 2> $O/ab_envv_$v.err || exit 1
  tail -n 1 $O/ab_envv_$v.out | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$VAR', '$v', 'value', round(j['value']), 'ms', j['ms_per_step'], 'parity', j.get('parity_vs_oracle'), 'rescore iso', (j['roofline'].get('rescore') or {}).get('kernel_ms_isolated'))"
done; done
