mkdir -p gpurun_out
python bench.py --shard lists > gpurun_out/bench_shard_n1.json 2> gpurun_out/bench_shard_n1.err
tail -3 gpurun_out/bench_shard_n1.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --steps 10 > gpurun_out/bench_shard_2rank_gloo.json 2> gpurun_out/bench_shard_2rank_gloo.err
tail -3 gpurun_out/bench_shard_2rank_gloo.err
python - <<'PY'
import json
for f in ("gpurun_out/bench_shard_n1.json","gpurun_out/bench_shard_2rank_gloo.json"):
    for l in open(f):
        if l.startswith("{"):
            j=json.loads(l); print(f, j["value"], j["ms_per_step"], json.dumps(j.get("list_sharded")))
PY
