R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
timeout -k 10 600 python scripts/r05_debug_onephase.py > $O/run2_debug.txt 2>&1; echo "debug rc=$?"; tail -40 $O/run2_debug.txt
timeout -k 10 900 python -m pytest tests/test_shard_gpu.py -q -m gpu -k "one_phase or simulated_peers" > $O/run2_tests.txt 2>&1; echo "tests rc=$?"; tail -15 $O/run2_tests.txt
