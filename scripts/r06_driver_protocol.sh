# round 6: the driver's own command, timed; stdout (detail line + compact final line) kept under gpurun_out/r06/
R=$PWD; O=$R/gpurun_out/r06; mkdir -p $O
t0=$(date +%s)
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.out 2> $O/bench_default.err
echo "rc=$? seconds=$(( $(date +%s) - t0 ))"; grep "bench\]" $O/bench_default.err | tail -40
tail -n 1 $O/bench_default.out > $O/bench_default.json
cp gpurun_out/bench/bench_detail.json $O/bench_default_detail.json
cp gpurun_out/bench/rocprofv3_kernel_stats_timed_region.csv $O/ 2>/dev/null
wc -c $O/bench_default.json
cat $O/bench_default.json
