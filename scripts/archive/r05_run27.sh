R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
run() { lab=$1; shift
  timeout -k 10 900 env "$@" > $O/run27_$lab.json 2> $O/run27_$lab.err
  python3 - $lab <<'PY'
import json, sys
lab = sys.argv[1]
try:
    j = json.loads([l for l in open(f"gpurun_out/r05/run27_{lab}.json") if l.startswith("{")][-1])
    print(lab, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in j.items() if k in ("ms_per_step", "unsharded_ms_per_step", "implied_strong_scaling_efficiency_without_links", "identical_rows_vs_replica", "rows")}, [round(x, 2) for x in j["windows_ms"]], j["scan"]["form"][:20])
except Exception as e:
    print(lab, "failed", repr(e)); print(open(f"gpurun_out/r05/run27_{lab}.err").read()[-800:])
PY
}
export GPU_MAX_HW_QUEUES=8
run lazy0 X=1 python3 scripts/r05_rank_share.py --workload c5 --depth 8 --co 12 --exchange dense --replay-lazy 0
run lazy1 X=1 python3 scripts/r05_rank_share.py --workload c5 --depth 8 --co 12 --exchange dense --replay-lazy 1
run lazy0b X=1 python3 scripts/r05_rank_share.py --workload c5 --depth 8 --co 12 --exchange dense --replay-lazy 0
run lazy1b X=1 python3 scripts/r05_rank_share.py --workload c5 --depth 8 --co 12 --exchange dense --replay-lazy 1
