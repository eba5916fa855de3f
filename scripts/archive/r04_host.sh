# round 4: is the host's enqueue time the limit?  (unprofiled)
mkdir -p gpurun_out/r04; O=gpurun_out/r04/host_enqueue.txt; : > $O
C="--steps 200 --warmup 10 --profile-only --shard none --traffic none --no-hbm-leg --no-cpu"
for i in 1 2; do
python bench.py $C 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('ms_per_step','host_enqueue_ms_per_step','ms_per_step_drained')})" >> $O
done
cat $O
