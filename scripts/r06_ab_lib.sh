# headline batch with two builds of the library in turn (TINYKNN_HIP_LIB): $A and $B, same box
O=gpurun_out/r06; mkdir -p $O
FLAGS="--steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --no-cpu --shard none $EXTRA"
for v in A B A B A B; do
  lib=$A; [ $v = B ] && lib=$B
  TINYKNN_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py $FLAGS > $O/ab_lib_$v.out 2> $O/ab_lib_$v.err || exit 1
  tail -n 1 $O/ab_lib_$v.out | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$v', '$lib', 'value', round(j['value']), 'ms', j['ms_per_step'], 'raw', j.get('raw_in_ids_out_queries_per_s'))"
done
