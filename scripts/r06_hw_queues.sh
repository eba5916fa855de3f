# headline batch by GPU_MAX_HW_QUEUES (bench.py sets 8 unless the environment says otherwise), same box, taken in turn
O=gpurun_out/r06; mkdir -p $O
B="--steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --no-cpu --shard none"
for v in 8 4 5 6 8 4 5 6; do
  GPU_MAX_HW_QUEUES=$v timeout -k 10 300 python bench.py $B > $O/hwq_$v.out 2> $O/hwq_$v.err || exit 1
  tail -n 1 $O/hwq_$v.out | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('GPU_MAX_HW_QUEUES', $v, 'value', round(j['value']), 'ms', j['ms_per_step'], 'raw', j.get('raw_in_ids_out_queries_per_s'), 'hipgraph', j.get('hipgraph_queries_per_s'))"
done
