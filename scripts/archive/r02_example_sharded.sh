#!/bin/bash
mkdir -p gpurun_out/filt
python examples/sharded.py --rows 1000000 2>&1 | grep -v amdgpu.ids | tail -3
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 \
   examples/sharded.py --rows 1000000 --backend gloo --exchange filtered 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -30
