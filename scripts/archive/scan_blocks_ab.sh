O=gpurun_out/scan_blocks; mkdir -p $O
for n in 256 384 448 512 576; do
  TINYKNN_SCAN_BLOCKS=$n python bench.py --profile-only --steps 200 --warmup 10 --shard none > $O/n$n.json 2> $O/n$n.err
done
python - <<'PY'
import json
for n in (256, 384, 448, 512, 576):
    j = json.load(open(f"gpurun_out/scan_blocks/n{n}.json"))
    print(f"{n}: pipelined {j['ms_per_step']:.3f} ms/step = {1e4 / j['ms_per_step'] / 1e3:.2f} M q/s; pipelined heap {j['stage_ms']['heap']:.3f} scan {j['stage_ms']['scan']:.3f} coarse_heap {j['stage_ms']['coarse_heap']:.3f}")
PY
