mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
tail -2 gpurun_out/bench_full.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --steps 10 > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err
tail -2 gpurun_out/bench_2rank.err
python - <<'PY'
import json
for f in ("gpurun_out/bench_full.json","gpurun_out/bench_2rank.json"):
    j=json.loads([l for l in open(f) if l.startswith("{")][0])
    print(f, round(j["value"]/1e6,2), round(j["ms_per_step"],3), round(j["roofline"]["frac"],3), j["parity_vs_oracle"], j.get("list_sharded"))
PY
