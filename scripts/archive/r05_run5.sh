R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
for d in 4 2 8; do
timeout -k 10 600 python3 scripts/r05_rank_share.py --steps 20 --depth $d > $O/run5_rank_share_d$d.json 2> $O/run5_rank_share_d$d.err
python3 - $d <<'PY'
import json, sys
d = sys.argv[1]
j = json.loads([l for l in open(f"gpurun_out/r05/run5_rank_share_d{d}.json") if l.startswith("{")][-1])
print("depth", d, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in j.items() if k in ("ms_per_step", "host_enqueue_ms_per_step", "unsharded_ms_per_step", "implied_strong_scaling_efficiency_without_links", "identical_rows_vs_replica")}, [round(x, 2) for x in j["windows_ms"]])
PY
done
