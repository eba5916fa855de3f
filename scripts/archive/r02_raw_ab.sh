# raw-in -> ids-out streaming session: copy engine (0) vs zero-copy kernels (1), with a timeline
R=$PWD; O=$R/gpurun_out/raw_ab; mkdir -p $O
python -m pytest tests/test_stream_gpu.py tests/test_full_size_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
for m in 0 1; do
  TINYKNN_STREAM_COPY=$m python scripts/raw_stream_probe.py > $O/raw_copy$m.json 2> $O/raw_copy$m.err
  TINYKNN_STREAM_COPY=$m python scripts/raw_stream_probe.py --prepared > $O/raw_copy${m}_prepared.json 2>> $O/raw_copy$m.err
done
TINYKNN_STREAM_COPY=1 python scripts/raw_stream_probe.py --slots 4 > $O/raw_copy1_slots4.json 2>> $O/raw_copy1.err
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  export TINYKNN_STREAM_COPY=$m
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/t$m -- python3 $R/scripts/raw_stream_probe.py --steps 60 > $O/trace$m.json 2> $O/trace$m.err
  python3 $R/scripts/trace_busy.py $O/t$m > $O/busy_copy$m.txt 2>&1
  rm -rf $O/t$m
done
cat $O/raw_copy*.json; tail -3 $O/pytest.log
