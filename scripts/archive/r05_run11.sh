R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
run() { # label, env..., args
  lab=$1; shift
  timeout -k 10 900 env "$@" > $O/run11_$lab.json 2> $O/run11_$lab.err
  python3 - $lab <<'PY'
import json, sys
lab = sys.argv[1]
try:
    j = json.loads([l for l in open(f"gpurun_out/r05/run11_{lab}.json") if l.startswith("{")][-1])
    print(lab, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in j.items() if k in ("ms_per_step", "host_enqueue_ms_per_step", "unsharded_ms_per_step", "implied_strong_scaling_efficiency_without_links", "identical_rows_vs_replica", "rows")}, [round(x, 2) for x in j["windows_ms"]], j["exchange"]["kind"][:8], j["scan"]["form"][:9])
    for r in j.get("stage_timeline_us", []):
        print("      ", r)
except Exception as e:
    print(lab, "failed", repr(e))
    print(open(f"gpurun_out/r05/run11_{lab}.err").read()[-1500:])
PY
}
run c5_d4 X=1 python3 scripts/r05_rank_share.py --workload c5 --depth 4
run c5_d8q8 GPU_MAX_HW_QUEUES=8 python3 scripts/r05_rank_share.py --workload c5 --depth 8
run g_roles2_q8 GPU_MAX_HW_QUEUES=8 TINYKNN_SHARD_ROLES=2 TINYKNN_SHARD_STAGE_EVENTS=1 python3 scripts/r05_rank_share.py --depth 4
run g_d8q8 GPU_MAX_HW_QUEUES=8 python3 scripts/r05_rank_share.py --depth 8
run g_d8q8_b GPU_MAX_HW_QUEUES=8 python3 scripts/r05_rank_share.py --depth 8
run g_d6q16 GPU_MAX_HW_QUEUES=16 python3 scripts/r05_rank_share.py --depth 6
