R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt4 -- python3 $R/scripts/r05_rank_share.py --steps 20 > $O/run4_rank_share.json 2> $O/run4_rank_share.err
echo "rc=$?"; cd $R
python3 scripts/r05_shard_trace.py $O/kt4 $O/run4_rank_share_trace.txt | head -60
rm -rf $O/kt4
tail -c 1500 $O/run4_rank_share.json
