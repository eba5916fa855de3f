R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
run() { # label, env..., args
  lab=$1; shift
  timeout -k 10 600 env "$@" > $O/run10_$lab.json 2> $O/run10_$lab.err
  python3 - $lab <<'PY'
import json, sys
lab = sys.argv[1]
try:
    j = json.loads([l for l in open(f"gpurun_out/r05/run10_{lab}.json") if l.startswith("{")][-1])
    print(lab, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in j.items() if k in ("ms_per_step", "host_enqueue_ms_per_step", "unsharded_ms_per_step", "implied_strong_scaling_efficiency_without_links", "identical_rows_vs_replica")}, [round(x, 2) for x in j["windows_ms"]])
    for r in j.get("stage_timeline_us", []):
        print("      ", r)
except Exception as e:
    print(lab, "failed", repr(e))
    print(open(f"gpurun_out/r05/run10_{lab}.err").read()[-1500:])
PY
}
run roles1 TINYKNN_SHARD_ROLES=1 TINYKNN_SHARD_STAGE_EVENTS=1 python3 scripts/r05_rank_share.py --depth 4
run roles2 TINYKNN_SHARD_ROLES=2 TINYKNN_SHARD_STAGE_EVENTS=1 python3 scripts/r05_rank_share.py --depth 4
run roles3 TINYKNN_SHARD_ROLES=3 python3 scripts/r05_rank_share.py --depth 6
run roles2d6 TINYKNN_SHARD_ROLES=2 python3 scripts/r05_rank_share.py --depth 6
