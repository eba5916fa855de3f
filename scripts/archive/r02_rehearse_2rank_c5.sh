#!/bin/bash
# BASELINE configs[4] (100M x 128, lists sharded) as two gloo ranks on the ONE GPU of the box: every
# rank generates and builds the index in HBM from the seed and keeps the codes of its lists
# (tk_index_shard_resident); dense and filtered exchange.  Rates mean nothing here (collectives
# staged through the host, two processes on one device): the rows must equal the unsharded index's.
mkdir -p gpurun_out/filt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
   bench.py --gpus 2 --backend gloo --workload c5 --steps 4 --warmup 2 --shard-depth 2 --shard-coalesce 1 \
   --shard-exchange both --no-cpu --no-hbm-leg --traffic none > gpurun_out/filt/two_rank_gloo_c5.json 2> gpurun_out/filt/two_rank_gloo_c5.err
echo "rc=$?"
grep "\[bench\]" gpurun_out/filt/two_rank_gloo_c5.err | tail -6
python - <<'PY'
import json
for l in open("gpurun_out/filt/two_rank_gloo_c5.json"):
    if l.startswith("{"):
        j = json.loads(l); ls = j["list_sharded"]
        print("value", round(j["value"]), j["scaling"], "rows", ls.get("identical_rows_vs_replica"), ls.get("error"),
              ls.get("exchange", {}).get("all_to_all_bytes_per_rank_per_step"), ls.get("code_chunks_per_rank"))
        f = ls.get("filtered_exchange")
        if f: print("filtered", round(f["queries_per_s"]), "rows", f["identical_rows_vs_replica"], f["exchange"]["bytes_ratio"],
                    f["exchange"]["record_bytes_per_rank_per_step"])
PY
