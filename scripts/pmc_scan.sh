# PMC passes on the list-major scan kernel (one batch in flight, so that the kernel runs alone).
# Counter names differ between ROCm releases: the list of the box is dumped first and every
# pass is its own rocprofv3 run (a refused name loses only that pass).
# (the TA_* counter set hangs the profiler on this pool - rc 124 / silence kill in both runs of
# round 2 - and is left out)
# usage: scripts/pmc_scan.sh <tag> [extra bench.py args]
R=$PWD
TAG=${1:-pmc}
shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
BENCH="python3 $R/bench.py --steps 4 --warmup 1 --no-cpu --shard none --recall-sample 10 --pipeline 1 --profile-only $*"
$BENCH > $O/warm.json 2> $O/warm.err      # builds + caches the index
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- $BENCH > /dev/null 2> $O/p$i.err
  echo "pass $i rc=$? : $set" >> $O/passes.txt
done <<'EOF'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum
TD_TD_BUSY_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
GRBM_GUI_ACTIVE GRBM_COUNT
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
EOF
cd $R
python3 - "$O" <<'PY'
import csv, glob, json, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open(O + "/pmc_summary.json", "w"), indent=1)
for k, d in out.items():
    if "scan_units" in k or "heap_replay_lanes" in k:
        print(k)
        for c, v in sorted(d.items()):
            print("   %-40s %14.1f  (%d)" % (c, v["mean"], v["launches"]))
PY
rm -rf $O/p[0-9]*/
