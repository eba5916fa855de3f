O=gpurun_out/r06; mkdir -p $O
B="--steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --no-cpu --shard none"
run() { tag=$1; shift; pl=$1; shift
  env "$@" timeout -k 10 300 python bench.py $B --pipeline $pl > $O/gs2_$tag.out 2> $O/gs2_$tag.err || { echo "$tag failed"; tail -3 $O/gs2_$tag.err; return; }
  python3 - $O/gs2_$tag.out $tag <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("# bench_detail ")][-1]
j = json.loads(l[len("# bench_detail "):])
g = j.get("hipgraph") or {}
print(sys.argv[2], "value", round(j["value"]), "graph", round(g.get("queries_per_s", 0)), "ratio", round(g.get("queries_per_s", 0) / j["value"], 3), "identical", g.get("identical_to_stream_launch"), g.get("error"))
PY
}
run gs3_p3 3 TINYKNN_GRAPH_STREAMS=3
run gs3_p4 4 TINYKNN_GRAPH_STREAMS=3
run gs3_p2_fq4 2 TINYKNN_GRAPH_STREAMS=3 DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run gs3_p3_fq4 3 TINYKNN_GRAPH_STREAMS=3 DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run gs4_p3 3 TINYKNN_GRAPH_STREAMS=4
