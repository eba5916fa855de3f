R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_shard_gpu.py -q -m gpu -x > $O/run16_tests.txt 2>&1; echo "tests rc=$?"; tail -5 $O/run16_tests.txt
run() { # label, env..., args
  lab=$1; shift
  timeout -k 10 900 env "$@" > $O/run16_$lab.json 2> $O/run16_$lab.err
  python3 - $lab <<'PY'
import json, sys
lab = sys.argv[1]
try:
    j = json.loads([l for l in open(f"gpurun_out/r05/run16_{lab}.json") if l.startswith("{")][-1])
    print(lab, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in j.items() if k in ("ms_per_step", "host_enqueue_ms_per_step", "unsharded_ms_per_step", "implied_strong_scaling_efficiency_without_links", "identical_rows_vs_replica", "rows", "steps_coalesced_per_exchange")}, [round(x, 2) for x in j["windows_ms"]], j["exchange"]["kind"][:8], j["scan"]["form"][:20])
except Exception as e:
    print(lab, "failed", repr(e))
    print(open(f"gpurun_out/r05/run16_{lab}.err").read()[-1500:])
PY
}
export GPU_MAX_HW_QUEUES=8
run c5_head X=1 python3 scripts/r05_rank_share.py --workload c5 --depth 8 --co 12 --exchange dense --plain 3
run c5_auto X=1 python3 scripts/r05_rank_share.py --workload c5 --depth 8 --co 12 --exchange dense
run g_head X=1 python3 scripts/r05_rank_share.py --depth 8 --co 12 --plain 3
run g_one X=1 python3 scripts/r05_rank_share.py --depth 8 --co 12
