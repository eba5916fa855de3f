# round 4: PMC passes over the plain-scan micro benchmark (wave-per-unit kernel); usage: r04_pmc_micro.sh OUTDIR K VARIANT
R=$PWD; O=$R/gpurun_out/r04/$1; mkdir -p $O
K=${2:-12}; V=${3:-1}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1 || true
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" \
           "TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -- $R/scripts/micro/bin/mfma_scan 1087 69 10000 9 512 0 $K $V > $O/run$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1); cp "$f" $O/pmc$i.csv 2>/dev/null; rm -rf $O/p$i
done
cd $R
python3 - $O <<'PY'
import csv, collections, sys
O = sys.argv[1]
for i in (1, 2, 3, 4, 5):
    try:
        rows = list(csv.DictReader(open(f"{O}/pmc{i}.csv")))
    except Exception as e:
        print("pass", i, "failed", e); continue
    acc = collections.defaultdict(list)
    for r in rows:
        if "scan_plain" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:32s} {sum(v)/len(v):16.0f}  ({len(v)} dispatches)")
PY
