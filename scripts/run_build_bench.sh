R=$PWD
mkdir -p gpurun_out/build
python bench_build.py > gpurun_out/build/bench_build.json 2> gpurun_out/build/bench_build.err
tail -2 gpurun_out/build/bench_build.err
cat gpurun_out/build/bench_build.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/build/kt -- python3 $R/bench_build.py --reps 1 --cpu-sample 1600 > /dev/null 2> $R/gpurun_out/build/kt.err
cd $R
f=$(find gpurun_out/build/kt -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/build/kernel_stats.csv; rm -rf gpurun_out/build/kt
grep -E "encode_pq|assign_kernel|normalise" gpurun_out/build/kernel_stats.csv | cut -c1-60,200-400
