R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
run() { # label, env..., args
  lab=$1; shift
  timeout -k 10 900 env "$@" > $O/run25_$lab.json 2> $O/run25_$lab.err
  python3 - $lab <<'PY'
import json, sys
lab = sys.argv[1]
try:
    j = json.loads([l for l in open(f"gpurun_out/r05/run25_{lab}.json") if l.startswith("{")][-1])
    print(lab, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in j.items() if k in ("ms_per_step", "host_enqueue_ms_per_step", "unsharded_ms_per_step", "implied_strong_scaling_efficiency_without_links", "identical_rows_vs_replica", "rows", "steps_coalesced_per_exchange")}, [round(x, 2) for x in j["windows_ms"]], j["exchange"]["kind"][:8], j["scan"]["form"][:20])
except Exception as e:
    print(lab, "failed", repr(e))
    print(open(f"gpurun_out/r05/run25_{lab}.err").read()[-1500:])
PY
}
export GPU_MAX_HW_QUEUES=8
run g_co8 X=1 python3 scripts/r05_rank_share.py --depth 8 --co 8
run g_co12 X=1 python3 scripts/r05_rank_share.py --depth 8 --co 12
run c5_co12 X=1 python3 scripts/r05_rank_share.py --workload c5 --depth 8 --co 12
run c5_co12_dense X=1 python3 scripts/r05_rank_share.py --workload c5 --depth 8 --co 12 --exchange dense
run c5_100M_co12_dense X=1 python3 scripts/r05_rank_share.py --workload c5 --n 100000000 --clusters 10000 --depth 8 --co 12 --exchange dense
