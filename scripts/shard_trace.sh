R=$PWD; O=$R/gpurun_out/shard_trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/bench.py --shard lists --shard-depth 4 --steps 40 --warmup 5 --no-cpu --no-hbm-leg --traffic none > $O/trace.json 2> $O/trace.err
f=$(find $O/t -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv
python3 $R/scripts/trace_busy.py $O/t 0.75 > $O/busy.txt 2>&1
rm -rf $O/t
head -30 $O/busy.txt
