# n_probes 20 / 50 (heaps of 211 / 511 entries: 63 / 140 KB of LDS per lane-replay wave): the replay launched whole
# (TINYKNN_REPLAY_PARTS=0) against in parts of a third of the chip's LDS room on two streams per replay stream, same box
O=gpurun_out/r06; mkdir -p $O
B="--steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --shard none"
for np in 50 20 10; do for v in 0 1 0 1; do
  TINYKNN_REPLAY_PARTS=$v timeout -k 10 300 python bench.py $B --n-probes $np > $O/parts_${np}_$v.out 2> $O/parts_${np}_$v.err || exit 1
  tail -n 1 $O/parts_${np}_$v.out | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('n_probes', $np, 'parts', $v, 'value', round(j['value']), 'ms', j['ms_per_step'], 'parity', j.get('parity_vs_oracle'))"
done; done
