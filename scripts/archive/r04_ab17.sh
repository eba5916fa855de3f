# round 4: list-sharded leg, streams by role (TINYKNN_SHARD_ROLES=1) against one stream per batch, with per-batch communicators
mkdir -p gpurun_out/r04; O=gpurun_out/r04/ab_shard_roles.txt; : > $O
C="--steps 100 --warmup 10 --traffic none --no-hbm-leg --no-cpu --sweep none --recall-sample 10"
for v in 0 1 0 1; do
  echo "== TINYKNN_SHARD_ROLES=$v" >> $O
  TINYKNN_SHARD_ROLES=$v python bench.py $C 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['list_sharded']; print('value', round(d['value']/1e6,2), 'sharded', round(r['queries_per_s']/1e6,2) if 'queries_per_s' in r else r, 'fixedQ', round(r.get('fixed_q_per_exchange',{}).get('queries_per_s',0)/1e6,2))" >> $O
done
cat $O
