R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
run() { lab=$1; shift
timeout -k 10 600 env "$@" python bench.py --steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --no-cpu --shard none > $O/run22_$lab.json 2> $O/run22_$lab.err
python3 - $lab <<'PY'
import json, sys
lab = sys.argv[1]
try:
    j = json.loads([l for l in open(f"gpurun_out/r05/run22_{lab}.json") if l.startswith("{")][-1])
    g = j.get("hipgraph") or {}
    print(lab, "value", round(j["value"]), "graph", round(g.get("queries_per_s", 0)), "ratio", round(g.get("queries_per_s", 0) / j["value"], 3), "same", g.get("identical_to_stream_launch"), g.get("error"))
except Exception as e:
    print(lab, "failed", repr(e)); print(open(f"gpurun_out/r05/run22_{lab}.err").read()[-800:])
PY
}
run base X=1
run q8 DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run q2 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run pc0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run pc1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run b1 DEBUG_HIP_GRAPH_BATCH_SIZE=1
run hwq4 GPU_MAX_HW_QUEUES=4
