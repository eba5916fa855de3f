# the headline index by queries per call (pipelined mode, pairs of calls where they fit): is there a cliff?
O=gpurun_out/r06; mkdir -p $O
FLAGS="--steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --shard none --cpu-sample 2000"
for nq in 500 1000 2000 5000 10000 20000 40000 80000; do
  timeout -k 10 300 python bench.py $FLAGS --nq $nq > $O/nq_$nq.out 2> $O/nq_$nq.err || exit 1
  tail -n 1 $O/nq_$nq.out | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('nq', $nq, 'value', round(j['value']), 'ms_per_call', j['ms_per_step'], 'parity', j.get('parity_vs_oracle'), 'tile_fill', j['roofline'].get('tile_fill'))"
done
