# round 4: kernel timeline of the one-rank list-sharded leg (dense exchange, forced RCCL)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04/shard_tl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -o tl -- python3 $R/bench.py --steps 40 --warmup 5 --shard lists --force-collectives --shard-exchange dense --no-cpu --sweep none --traffic none --no-hbm-leg --recall-sample 10 "$@" > $O/bench.json 2> $O/bench.err
ls -la $O
