# round 4: batches in flight of the list-sharded leg (one stream each + RCCL's internal streams: HIP has four hardware queues)
mkdir -p gpurun_out/r04; O=gpurun_out/r04/ab_shard_depth.txt; : > $O
C="--steps 100 --warmup 10 --traffic none --no-hbm-leg --no-cpu --sweep none --recall-sample 10"
for v in "4 1" "2 1" "3 1" "2 0" "3 0" "4 1"; do
  set -- $v
  echo "== --shard-depth $1 TINYKNN_SHARD_COMMS=$2" >> $O
  TINYKNN_SHARD_COMMS=$2 python bench.py $C --shard-depth $1 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['list_sharded']; print('value', round(d['value']/1e6,2), 'sharded', round(r['queries_per_s']/1e6,2) if 'queries_per_s' in r else r, 'filtered', round(r.get('filtered_exchange',{}).get('queries_per_s',0)/1e6,2), 'fixedQ', round(r.get('fixed_q_per_exchange',{}).get('queries_per_s',0)/1e6,2))" >> $O
done
cat $O
