R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_shard_gpu.py tests/test_stream_gpu.py -q -m gpu -x > $O/run13_tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/run13_tests.txt
for cfg in "3 4" "3 8" "4 8" "6 8" "4 6" "6 12"; do
set -- $cfg
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --sweep none --traffic none --no-hbm-leg --no-cpu --rank-share 0 --shard-exchange dense --shard-coalesce $1 --shard-depth $2 > $O/run13_co$1_d$2.json 2> $O/run13_co$1_d$2.err
python3 - $1 $2 <<'PY'
import json, sys
co, d = sys.argv[1:3]
j = json.loads([l for l in open(f"gpurun_out/r05/run13_co{co}_d{d}.json") if l.startswith("{")][-1])
ls = j.get("list_sharded", {})
print("co", co, "depth", d, "value", round(j["value"]), "sharded", round(ls.get("queries_per_s", 0)), "ratio", round(ls.get("ratio_to_unsharded_value", 0), 3), "fixedq", round(ls.get("fixed_q_per_exchange", {}).get("queries_per_s", 0)),
      "windows", [round(x, 2) for x in ls.get("windows_ms", [])], "rep", ls.get("windows_repeated_after_overflow"))
PY
done
