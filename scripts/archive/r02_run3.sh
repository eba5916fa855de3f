# round 2, run 3: the whole GPU suite, the default bench line, the 2-rank rehearsal of the N > 1 path
R=$PWD; O=$R/gpurun_out/r02_run3; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?"
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --steps 30 --warmup 3 --no-cpu > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; echo "2rank rc=$?"
python - <<'PY'
import json
for f in ("bench_full", "bench_2rank_gloo"):
    try:
        j = json.loads([l for l in open(f"gpurun_out/r02_run3/{f}.json") if l.startswith("{")][-1])
    except Exception as e:
        print(f, "FAILED", e); continue
    print(f, j["value"], j["ms_per_step"], j["scaling"], j["n_gpus"], j.get("raw_in_ids_out", {}) and j["raw_in_ids_out"].get("queries_per_s"),
          j.get("list_sharded"), j.get("replica", {}).get("queries_per_s"))
PY
