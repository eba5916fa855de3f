R=$PWD
mkdir -p gpurun_out/trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace/p -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu --shard none --recall-sample 10 > $R/gpurun_out/trace/p.json 2> $R/gpurun_out/trace/p.err
cd $R
f=$(find gpurun_out/trace/p -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sc = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "scan_units2" in r["Kernel_Name"]]
gaps = [(sc[i+1][0]-sc[i][1])/1e3 for i in range(len(sc)-1)]
durs = [(b-a)/1e3 for a,b in sc]
print("fused launches", len(sc))
print("durations us:", [round(x) for x in durs[8:30]])
print("gaps us:", [round(x) for x in gaps[8:30]])
# what runs in a typical gap: kernels overlapping [end_i, start_{i+1}] for i=15
i=15
lo,hi=sc[i][1],sc[i+1][0]
for r in rows:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if e>lo-50000 and s<hi+50000:
        print(r["Kernel_Name"][:45], r["Stream_Id"], round((s-lo)/1e3), round((e-lo)/1e3))
PY
rm -rf gpurun_out/trace/p
