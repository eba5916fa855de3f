# n_probes 20 / 50 in the pipelined batch (heaps of 211 / 511 entries): lane replay (heap mode 0) against the register heap
# with four / eight nodes per lane forced for both replays (heap mode 3), same box, taken in turn
O=gpurun_out/r06; mkdir -p $O
B="--steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --no-cpu --shard none"
for np in 20 50; do for hm in 0 3 0 3; do
  timeout -k 10 300 python bench.py $B --n-probes $np --heap-mode $hm > $O/ab_np_${np}_$hm.out 2> $O/ab_np_${np}_$hm.err || exit 1
  tail -n 1 $O/ab_np_${np}_$hm.out | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('n_probes', $np, 'heap_mode', $hm, 'value', round(j['value']), 'ms', j['ms_per_step'], 'parity', j.get('parity_vs_oracle'))"
done; done
