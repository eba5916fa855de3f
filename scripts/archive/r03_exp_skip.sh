# round 3 timing experiment (WRONG results on purpose; library built with -DTK_TIMING_EXPERIMENTS): what does the pipelined
# batch cost without one of its kernels?  1 = no list replay, 2 = no final rescoring, 16 = no plain kernel
# build the experiment library first (it never ships):
#   for f in adc_scan plain_scan heap tables rescore shard build brute front devbuild api; do \
#     hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DTK_TIMING_EXPERIMENTS \
#           -c tinyknn_amd/csrc/$f.hip -o /tmp/expbuild/$f.o; done
#   hipcc --offload-arch=gfx950 -shared -fPIC -o tinyknn_amd/libtinyknn_hip_exp.so /tmp/expbuild/*.o -ldl -lpthread
R=$PWD; O=$R/gpurun_out/r03_exp; mkdir -p $O
export TINYKNN_HIP_LIB=$R/tinyknn_amd/libtinyknn_hip_exp.so
for sk in 0 1 2 16 3 18 19; do
  TINYKNN_DEBUG_SKIP=$sk python bench.py --shard none --traffic none --profile-only --steps 50 > $O/skip$sk.json 2> $O/skip$sk.err
  python3 - $O/skip$sk.json $sk <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = lambda d: {k: round(v, 3) for k, v in d.items()}
    print("skip", sys.argv[2].ljust(4), "ms", round(j["ms_per_step"], 3), r(j["stage_ms"]), flush=True)
except Exception as e:
    print("skip", sys.argv[2], "failed", e, flush=True)
PY
done
