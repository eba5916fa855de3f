# one rank's share of W = 8 (GloVe-shaped), the home queries' replay by the lane kernel (TINYKNN_PAIR_NQ=2048) against the
# wave-per-query register heap (20000), same box, in turn
O=gpurun_out/r06; mkdir -p $O
for v in 2048 20000 2048 20000; do
  TINYKNN_PAIR_NQ=$v GPU_MAX_HW_QUEUES=8 timeout -k 10 300 python scripts/r05_rank_share.py --depth 8 > $O/rs_pair_$v.out 2> $O/rs_pair_$v.err || { tail -5 $O/rs_pair_$v.err; exit 1; }
  tail -n 1 $O/rs_pair_$v.out | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('pair_nq', $v, 'rank ms/step', round(j['ms_per_step'],4), 'unsharded', round(j['unsharded_ms_per_step'],4), 'eff', round(j['implied_strong_scaling_efficiency_without_links'],3), 'rows', j['identical_rows_vs_replica'], j['rows'])"
done
