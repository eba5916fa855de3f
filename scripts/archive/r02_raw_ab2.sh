R=$PWD; O=$R/gpurun_out/raw_ab2; mkdir -p $O
python -m pytest tests/test_stream_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
for m in 0 1; do
  TINYKNN_STREAM_COPY=$m python scripts/raw_stream_probe.py > $O/raw_copy$m.json 2> $O/raw_copy$m.err
done
TINYKNN_STREAM_OWN_STREAM=1 python scripts/raw_stream_probe.py > $O/raw_copy1_ownstream.json 2>> $O/raw_copy1.err
python scripts/raw_stream_probe.py --slots 4 > $O/raw_copy1_slots4.json 2>> $O/raw_copy1.err
python scripts/raw_stream_probe.py --slots 16 > $O/raw_copy1_slots16.json 2>> $O/raw_copy1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/t1 -- python3 $R/scripts/raw_stream_probe.py --steps 60 > $O/trace1.json 2> $O/trace1.err
python3 $R/scripts/trace_busy.py $O/t1 > $O/busy_copy1.txt 2>&1
rm -rf $O/t1
cat $O/raw_copy*.json; tail -3 $O/pytest.log; head -8 $O/busy_copy1.txt; tail -1 $O/busy_copy1.txt
