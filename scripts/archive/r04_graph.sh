# round 4: the pipelined mode as ONE hipGraph: replay time against the number of captured steps (fill + drain once per replay)
mkdir -p gpurun_out/r04; O=gpurun_out/r04/graph_steps.txt; : > $O
C="--steps 100 --warmup 10 --shard none --traffic none --no-hbm-leg --no-cpu --sweep none --recall-sample 10"
for g in 8 32 96; do
  echo "== --graph-steps $g" >> $O
  python bench.py $C --graph-steps $g 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); g=d['hipgraph']; print('value', round(d['value']/1e6,2), 'graph ms_per_replay', g.get('ms_per_replay'), 'ms_per_step', g.get('ms_per_step'), 'M q/s', round(g.get('queries_per_s',0)/1e6,2), g.get('identical_to_stream_launch'), g.get('error'))" >> $O
done
cat $O
