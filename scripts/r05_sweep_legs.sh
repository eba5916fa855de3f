# which leg in front of the sweep lowers its build(n_probes=2) point?  (full line: 13.1 M; without the traffic, HBM-scale and CPU legs: 14.9-15.1 M)
R=$PWD; O=$R/gpurun_out/r05b; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() { lab=$1; shift
  timeout -k 10 500 python3 $R/bench.py --steps 20 --warmup 5 --rank-share 0 --shard none --recall-sample 10 "$@" > $O/legs_$lab.json 2> $O/legs_$lab.err
  python3 - $lab <<'PY'
import json, sys
lab = sys.argv[1]
j = json.loads([l for l in open(f"/root/repo/gpurun_out/r05b/legs_{lab}.json") if l.startswith("{")][-1])
print(lab, "value", round(j["value"] / 1e6, 2), [round(p.get("queries_per_s", 0) / 1e6, 2) for p in j["sweep"]["points"]])
PY
}
run hbm_on --traffic none
run traffic_on --no-hbm-leg
