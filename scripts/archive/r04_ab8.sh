# round 4: one RCCL communicator per batch in flight (TINYKNN_SHARD_COMMS=1, default) against one for all (0):
# the one-rank list-sharded rehearsal of the default bench line
mkdir -p gpurun_out/r04; O=gpurun_out/r04/ab_shard_comms.txt; : > $O
C="--steps 100 --warmup 10 --traffic none --no-hbm-leg --no-cpu --sweep none --recall-sample 10"
for v in 0 1 0 1; do
  echo "== TINYKNN_SHARD_COMMS=$v" >> $O
  TINYKNN_SHARD_COMMS=$v python bench.py $C 2>gpurun_out/r04/ab8.err | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['list_sharded']; print('value', round(d['value']/1e6,2), 'sharded', round(r['queries_per_s']/1e6,2) if 'queries_per_s' in r else r, 'filtered', round(r.get('filtered_exchange',{}).get('queries_per_s',0)/1e6,2), 'fixedQ', round(r.get('fixed_q_per_exchange',{}).get('queries_per_s',0)/1e6,2), 'identical', r.get('identical_rows_vs_replica'))" >> $O
  tail -2 gpurun_out/r04/ab8.err >> $O
done
cat $O
