# one rank's share of W = 8 (GloVe-shaped, dense exchange): tables built at home and gathered against tables on every rank
R=$PWD; O=$R/gpurun_out/r05b; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd /tmp; export TMPDIR=/tmp
for i in 1 2; do for t in all home; do
  timeout -k 10 300 python3 $R/scripts/r05_rank_share.py --depth 8 --co 8 --tables $t > $O/rank_tables_$t$i.json 2> $O/rank_tables_$t$i.err
  python3 - $t $i <<'PY'
import json, sys
t, i = sys.argv[1], sys.argv[2]
try:
    j = json.loads([l for l in open(f"/root/repo/gpurun_out/r05b/rank_tables_{t}{i}.json") if l.startswith("{")][-1])
    print(t, i, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in j.items() if k in ("ms_per_step", "unsharded_ms_per_step", "implied_strong_scaling_efficiency_without_links", "identical_rows_vs_replica", "rows", "tables")})
except Exception as e:
    print(t, i, "failed", repr(e)); print(open(f"/root/repo/gpurun_out/r05b/rank_tables_{t}{i}.err").read()[-800:])
PY
done; done
