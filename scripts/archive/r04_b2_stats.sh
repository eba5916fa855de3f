# round 4: kernel stats of the build(n_probes=2) configuration (labels repeat: duplicate-test replay)
mkdir -p gpurun_out/r04
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04/b2 -o b2 -- python3 $R/bench.py --steps 50 --warmup 5 --profile-only --shard none --traffic none --no-hbm-leg --no-cpu --build-probes 2 > $R/gpurun_out/r04/b2_stdout.txt 2>&1
ls $R/gpurun_out/r04/b2
