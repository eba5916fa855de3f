# small pipelined calls (500 ... 4 000 queries per call, pairs of calls per launch): the register heap's limit with batches in flight,
# 256 (until round 6) against 8 192, in turn (TINYKNN_PAIR_NQ_PIPE); the default became 4 096 per launch
O=gpurun_out/r06; mkdir -p $O
FLAGS="--steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --shard none --cpu-sample 2000"
for nq in 500 1000 2000 4000; do for cap in 256 8192 256 8192; do
  TINYKNN_PAIR_NQ_PIPE=$cap timeout -k 10 300 python bench.py $FLAGS --nq $nq > $O/nqp.out 2> $O/nqp.err || exit 1
  tail -n 1 $O/nqp.out | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('nq', $nq, 'cap', $cap, 'value', round(j['value']), 'ms_per_call', j['ms_per_step'], 'parity', j.get('parity_vs_oracle'))"
done; done
