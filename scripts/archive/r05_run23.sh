R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt23 -- python3 $R/scripts/r05_graph_trace.py > $O/run23_graph.txt 2> $O/run23_graph.err
echo "rc=$?"; cd $R; grep -E "stream-launched|graph" $O/run23_graph.txt
python3 scripts/r05_graph_trace_read.py $O/kt23 | tee $O/run23_graph_trace.txt
rm -rf $O/kt23
