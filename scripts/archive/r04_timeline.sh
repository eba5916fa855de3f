# round 4: kernel timeline of the timed region (who overlaps whom, where the chip idles)
mkdir -p gpurun_out/r04
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04/tl -o tl -- python3 $R/bench.py --steps 50 --warmup 5 --profile-only --shard none --traffic none --no-hbm-leg --no-cpu "$@" > $R/gpurun_out/r04/tl_stdout.txt 2>&1
ls -la $R/gpurun_out/r04/tl
