# hipGraph of the pipelined mode captured on 4 (as stream-launched) / 3 / 2 streams: the executor replays on two queues anyway
O=gpurun_out/r06; mkdir -p $O
B="--steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --no-cpu --shard none"
for v in 4 2 3 2; do
  TINYKNN_GRAPH_STREAMS=$v timeout -k 10 300 python bench.py $B > $O/gs_$v.out 2> $O/gs_$v.err || { echo "$v failed"; tail -3 $O/gs_$v.err; continue; }
  python3 - $O/gs_$v.out $v <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("# bench_detail ")][-1]
j = json.loads(l[len("# bench_detail "):])
g = j.get("hipgraph") or {}
print("graph streams", sys.argv[2], "value", round(j["value"]), "graph", round(g.get("queries_per_s", 0)), "ratio", round(g.get("queries_per_s", 0) / j["value"], 3), "identical", g.get("identical_to_stream_launch"), g.get("error"), "one batch", (g.get("one_batch_in_flight") or {}).get("queries_per_s"))
PY
done
