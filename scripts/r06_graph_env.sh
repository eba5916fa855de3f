# hipGraph of the pipelined mode under the runtime's remaining knobs (round 5 tried DEBUG_HIP_FORCE_GRAPH_QUEUES 2 / 8,
# DEBUG_CLR_GRAPH_PACKET_CAPTURE, DEBUG_HIP_GRAPH_BATCH_SIZE, GPU_MAX_HW_QUEUES 4 / 8), and what the executor says it does
O=gpurun_out/r06; mkdir -p $O
B="--steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --no-cpu --shard none"
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py $B > $O/graph_$tag.out 2> $O/graph_$tag.err || { echo "$tag failed"; tail -3 $O/graph_$tag.err; return; }
  python3 - $O/graph_$tag.out $tag <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("# bench_detail ")][-1]
j = json.loads(l[len("# bench_detail "):])
g = j.get("hipgraph") or {}
print(sys.argv[2], "value", round(j["value"]), "graph", round(g.get("queries_per_s", 0)), "ratio", round(g.get("queries_per_s", 0) / j["value"], 3), "identical", g.get("identical_to_stream_launch"), g.get("error"))
PY
}
run base X=1
run dynq0 DEBUG_HIP_DYNAMIC_QUEUES=0
run dynq1 DEBUG_HIP_DYNAMIC_QUEUES=1
run asyncq DEBUG_HIP_FORCE_ASYNC_QUEUE=1
run fq4 DEBUG_HIP_FORCE_GRAPH_QUEUES=4
AMD_LOG_LEVEL=4 timeout -k 10 300 python bench.py $B > $O/graph_log.out 2> $O/graph_log.err; grep -i "hipGraph\]\|GraphExec::Run\|max streams" $O/graph_log.err | sort | uniq -c | sort -rn | head -20 > $O/graph_log_summary.txt; cat $O/graph_log_summary.txt; rm -f $O/graph_log.err
