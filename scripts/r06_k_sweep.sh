# the headline index by k (heaps of (n_probes + 1) k + 1 entries: 12 / 56 / 111 / 221 / 551 / 1101), pipelined, 10 000 queries per call
O=gpurun_out/r06; mkdir -p $O
FLAGS="--steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --shard none --cpu-sample 2000"
for k in 1 5 10 20 50 100; do
  timeout -k 10 300 python bench.py $FLAGS --k $k > $O/k_$k.out 2> $O/k_$k.err || { tail -5 $O/k_$k.err; exit 1; }
  tail -n 1 $O/k_$k.out | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('k', $k, 'value', round(j['value']), 'ms_per_call', j['ms_per_step'], 'parity', j.get('parity_vs_oracle'), 'cpu', round(j['cpu_baseline']['value']))"
done
