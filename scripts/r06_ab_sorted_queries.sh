# headline batch with the queries of every batch ordered by their nearest coarse centre (none / probe / probe-xcd), same box, in turn
O=gpurun_out/r06; mkdir -p $O
B="--steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --shard none --cpu-sample 1000"
for v in none probe probe-xcd none probe probe-xcd; do
  timeout -k 10 300 python bench.py $B --sort-queries $v > $O/ab_sort_$v.out 2> $O/ab_sort_$v.err || { tail -3 $O/ab_sort_$v.err; exit 1; }
  python3 - $O/ab_sort_$v.out $v <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("# bench_detail ")][-1]
j = json.loads(l[len("# bench_detail "):])
print(sys.argv[2], "value", round(j["value"]), "ms", round(j["ms_per_step"], 4), "parity", j["parity_vs_oracle"], "stage_ms", {k: round(v, 3) for k, v in j["stage_ms"].items()}, "iso", {k: round(v, 3) for k, v in j["isolated"]["stage_ms"].items()})
PY
done
