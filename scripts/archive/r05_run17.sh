R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt17 -- python3 $R/scripts/r05_rank_share.py --workload c5 --depth 8 --co 12 --exchange dense --plain 3 > $O/run17_c5.json 2> $O/run17_c5.err
echo "rc=$?"; cd $R
python3 scripts/r05_shard_trace.py $O/kt17 $O/run17_c5_trace.txt | head -48
rm -rf $O/kt17
