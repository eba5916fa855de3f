# headline batch with the wave-per-query register heap in place of the lane replay (both replays / never), same box
O=gpurun_out/r06; mkdir -p $O
B="--steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --no-cpu --shard none"
for v in 256 1000000 256 1000000; do
  TINYKNN_PAIR_NQ=$v timeout -k 10 300 python bench.py $B > $O/ab_pair_$v.out 2> $O/ab_pair_$v.err || exit 1
  tail -n 1 $O/ab_pair_$v.out | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('pair_nq', $v, 'value', round(j['value']), 'ms', j['ms_per_step'], 'parity', j['parity_vs_oracle'])"
done
