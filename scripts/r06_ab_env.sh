# headline batch with an environment switch of the library off / on in turn ($VAR=0 / 1), same box; $EXTRA = more bench flags
O=gpurun_out/r06; mkdir -p $O
FLAGS="--steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --shard none $EXTRA"
for v in 0 1 0 1 0 1 0 1; do
  env $VAR=$v timeout -k 10 300 python bench.py $FLAGS > $O/ab_env_$v.out 2> $O/ab_env_$v.err || exit 1
  tail -n 1 $O/ab_env_$v.out | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$VAR', $v, 'value', round(j['value']), 'ms', j['ms_per_step'], 'parity', j.get('parity_vs_oracle'))"
done
