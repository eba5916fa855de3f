R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt15 -- python3 $R/scripts/r05_rank_share.py --workload c5 --depth 8 --co 12 > $O/run15_c5.json 2> $O/run15_c5.err
echo "rc=$?"; cd $R
python3 scripts/r05_shard_trace.py $O/kt15 $O/run15_c5_trace.txt | head -60
rm -rf $O/kt15
