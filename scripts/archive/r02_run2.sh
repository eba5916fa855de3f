# round 2, run 2: VALU issue-rate microbench, GPU tests, A/B of the scan forms, full bench line
R=$PWD; O=$R/gpurun_out/r02_run2; mkdir -p $O
scripts/micro/bin/valu_rate > $O/valu_rate.txt 2>&1
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
for f in 0 1 2; do
  python bench.py --steps 100 --warmup 10 --no-cpu --traffic none --no-hbm-leg --shard none --recall-sample 10 --scan-form $f > $O/bench_form$f.json 2> $O/bench_form$f.err
done
python bench.py > $O/bench_full.json 2> $O/bench_full.err
