R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
run() { # label, env..., args
  lab=$1; shift
  timeout -k 10 600 env "$@" > $O/run8_$lab.json 2> $O/run8_$lab.err
  python3 - $lab <<'PY'
import json, sys
lab = sys.argv[1]
try:
    j = json.loads([l for l in open(f"gpurun_out/r05/run8_{lab}.json") if l.startswith("{")][-1])
    print(lab, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in j.items() if k in ("ms_per_step", "host_enqueue_ms_per_step", "unsharded_ms_per_step", "implied_strong_scaling_efficiency_without_links", "identical_rows_vs_replica")}, [round(x, 2) for x in j["windows_ms"]])
except Exception as e:
    print(lab, "failed", repr(e))
PY
}
run d4 X=1 python3 scripts/r05_rank_share.py --depth 4
run d8 X=1 python3 scripts/r05_rank_share.py --depth 8
run d8q8 GPU_MAX_HW_QUEUES=8 python3 scripts/r05_rank_share.py --depth 8
run d12q8 GPU_MAX_HW_QUEUES=8 python3 scripts/r05_rank_share.py --depth 12
run roles4 TINYKNN_SHARD_ROLES=1 python3 scripts/r05_rank_share.py --depth 4
run roles6 TINYKNN_SHARD_ROLES=1 python3 scripts/r05_rank_share.py --depth 6
