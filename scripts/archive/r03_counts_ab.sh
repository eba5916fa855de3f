#!/bin/bash
# round 3: filtered exchange of the list-sharded leg at W = 1 — record counts read on the device (fixed regions,
# equal-split all-to-all, no host synchronisation) vs on the host (variable splits); glove-like and c5
O=gpurun_out/r03_counts; mkdir -p $O
show() { python3 - $1 $2 <<'P'
import json, sys
l = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
ls = l["list_sharded"]
print(sys.argv[2], "unsharded", round(l["value"]), "| dense", round(ls["queries_per_s"]), ls["identical_rows_vs_replica"], flush=True)
f = ls.get("filtered_exchange")
if f: print("    filtered", round(f["queries_per_s"]), f["identical_rows_vs_replica"], {k: v for k, v in f["exchange"].items() if k in ("bytes_ratio", "host_syncs_per_exchange", "record_region", "record_bytes_per_rank_per_step", "records_held_bytes_per_rank_per_step", "whole_segment_bytes_per_rank_per_step")}, flush=True)
P
}
for c in device host; do
  python bench.py --shard lists --shard-exchange both --shard-counts $c --no-cpu --no-hbm-leg --traffic none --steps 30 --warmup 6 > $O/glove_$c.json 2> $O/glove_$c.log && show $O/glove_$c.json glove_$c || exit 1
done
for c in device host; do
  python bench.py --workload c5 --shard lists --shard-exchange both --shard-counts $c --shard-coalesce 1 --shard-depth 2 \
     --no-cpu --no-hbm-leg --traffic none --steps 30 --warmup 4 > $O/c5_$c.json 2> $O/c5_$c.log && show $O/c5_$c.json c5_$c || exit 1
done
