#!/bin/bash
# round 3: list-sharded leg at W = 1, two-phase scan (matrix cores behind the first lists) vs exact kernel only
O=gpurun_out/r03_shard_plain; mkdir -p $O
show() { python3 - $1 $2 <<'P'
import json, sys
l = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
ls = l["list_sharded"]
print(sys.argv[2], "unsharded", round(l["value"]), "| dense", round(ls["queries_per_s"]), ls["identical_rows_vs_replica"], "fixed-Q", ls.get("fixed_q_per_exchange", {}).get("queries_per_s") and round(ls["fixed_q_per_exchange"]["queries_per_s"]), ls["scan"].get("last_batch_this_rank"), flush=True)
f = ls.get("filtered_exchange")
if f: print("    filtered", round(f["queries_per_s"]), f["identical_rows_vs_replica"], flush=True)
P
}
for pl in 1 0; do
  python bench.py --shard lists --shard-exchange both --shard-plain $pl --no-cpu --no-hbm-leg --traffic none --steps 30 --warmup 6 > $O/glove_plain$pl.json 2> $O/glove_plain$pl.log && show $O/glove_plain$pl.json glove_plain$pl || exit 1
done
for pl in 1 0; do
  python bench.py --workload c5 --shard lists --shard-exchange both --shard-plain $pl --shard-coalesce 1 --shard-depth 2 \
     --no-cpu --no-hbm-leg --traffic none --steps 30 --warmup 4 > $O/c5_plain$pl.json 2> $O/c5_plain$pl.log && show $O/c5_plain$pl.json c5_plain$pl || exit 1
done
