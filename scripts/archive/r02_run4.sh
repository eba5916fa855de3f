# round 2, run 4: GPU suite, default bench line (with the VALU fraction), configs[2] flat top() timing,
# per-stream busy fractions of the pipelined mode
R=$PWD; O=$R/gpurun_out/r02_run4; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?"
python scripts/time_flat_top.py > $O/flat_top_1m.json 2> $O/flat_top_1m.err; echo "flat rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/t -- python3 $R/bench.py --steps 60 --warmup 5 --profile-only --shard none > $O/trace.json 2> $O/trace.err
python3 $R/scripts/trace_busy.py $O/t > $O/busy_pipelined.txt 2>&1
rm -rf $O/t
cd $R
cat $O/flat_top_1m.json; head -12 $O/busy_pipelined.txt; tail -1 $O/busy_pipelined.txt
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r02_run4/bench_full.json") if l.startswith("{")][-1])
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["valu"], j["raw_in_ids_out"]["queries_per_s"], j["parity_vs_oracle"])
PY
