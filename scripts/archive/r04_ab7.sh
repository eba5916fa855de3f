# round 4: where the raw queries enter (experiment build): 0 kernel on the front stream, 1 kernel on the caller's (scan) stream,
# 2 kernel on an own stream, 3 copy engine on an own stream
mkdir -p gpurun_out/r04; O=gpurun_out/r04/ab_ingest.txt; : > $O
C="--steps 200 --warmup 10 --shard none --traffic none --no-hbm-leg --no-cpu --sweep none --recall-sample 10"
for v in 0 1 2 3 0 1; do
  echo "== TINYKNN_INGEST=$v" >> $O
  TINYKNN_INGEST=$v python bench.py $C 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['raw_in_ids_out']; print('value', round(d['value']/1e6,2), 'raw_in_ids_out', round(r['queries_per_s']/1e6,2), 'ms', round(r['ms_per_step'],4), 'identical', r['rows_identical_to_device_resident_path'])" >> $O
done
cat $O
