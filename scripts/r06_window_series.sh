# the timed region's windows over a long run (--windows 250: 5 000 steps, ~2 s), three processes in turn: how often and for how long
# the pipeline sits in its slower phase
O=gpurun_out/r06; mkdir -p $O
FLAGS="--steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --no-cpu --shard none --windows 250"
for r in 1 2 3; do
  timeout -k 10 300 python bench.py $FLAGS > $O/windows_$r.out 2> $O/windows_$r.err || exit 1
  python3 - $O/windows_$r.out <<'PY'
import json, sys
lines = open(sys.argv[1]).read().splitlines()
d = json.loads([l for l in lines if l.startswith("# bench_detail ")][-1][len("# bench_detail "):])
w = d["timing"]["window_ms"][1:-1]
med = sorted(w)[len(w) // 2]
slow = [x > 1.05 * med for x in w]
runs, cur = [], 0
for s in slow:
    if s: cur += 1
    elif cur: runs.append(cur); cur = 0
if cur: runs.append(cur)
print("value", round(d["value"]), "median window", round(med, 3), "min", round(min(w), 3), "max", round(max(w), 3),
      "windows > 1.05 x median:", sum(slow), "of", len(w), "in runs of", runs)
print("  series (ms, rounded):", " ".join("%.1f" % x for x in w))
PY
done
