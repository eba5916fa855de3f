O=gpurun_out/c5_ab2; mkdir -p $O
for g in 576 640 704; do for m in 0 1; do
  TINYKNN_SCAN_BLOCKS=$g TINYKNN_MERGE_EVENTS=$m python bench.py --workload c5 --profile-only --steps 60 --warmup 5 > $O/g${g}_m$m.json 2> $O/g${g}_m$m.err
done; done
python - <<'PY'
import json
for g in (576, 640, 704):
    for m in (0, 1):
        j = json.load(open(f"gpurun_out/c5_ab2/g{g}_m{m}.json"))
        print(f"grid {g} merge {m}: {j['ms_per_step']:.3f} ms/step = {1e4 / j['ms_per_step'] / 1e3:.2f} M q/s; scan {j['stage_ms']['scan']:.3f} heap {j['stage_ms']['heap']:.3f}")
PY
