# one rank's share of W = 8 (GloVe-shaped): batches in flight x hardware queues
O=gpurun_out/r06; mkdir -p $O
for cfg in "4 8" "6 8" "8 8" "8 16" "12 16" "16 24"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$2 timeout -k 10 300 python scripts/r05_rank_share.py --depth $1 > $O/rs_d$1_q$2.out 2> $O/rs_d$1_q$2.err || { tail -3 $O/rs_d$1_q$2.err; continue; }
  tail -n 1 $O/rs_d$1_q$2.out | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('depth', $1, 'hwq', $2, 'rank ms/step', round(j['ms_per_step'],4), 'eff', round(j['implied_strong_scaling_efficiency_without_links'],3), 'host enqueue', round(j['host_enqueue_ms_per_step'],4), 'rows', j['identical_rows_vs_replica'])"
done
