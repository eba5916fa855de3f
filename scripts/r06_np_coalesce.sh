# n_probes 20 / 50 (heaps of 211 / 511 entries: 54 / 131 KB of LDS per lane-replay wave): pairs of calls in one launch
# (314 waves) against single calls (157 waves per launch, two replay streams), same box
O=gpurun_out/r06; mkdir -p $O
B="--steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --no-cpu --shard none"
for np in 20 50; do for co in 2 1 2 1; do
  timeout -k 10 300 python bench.py $B --n-probes $np --coalesce $co > $O/np_co_${np}_$co.out 2> $O/np_co_${np}_$co.err || exit 1
  tail -n 1 $O/np_co_${np}_$co.out | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('n_probes', $np, 'coalesce', $co, 'value', round(j['value']), 'ms', j['ms_per_step'])"
done; done
