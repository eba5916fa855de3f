R=$PWD
mkdir -p gpurun_out/trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace/p3 -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu --shard none > $R/gpurun_out/trace/p3.json 2> $R/gpurun_out/trace/p3.err
cd $R
f=$(find gpurun_out/trace/p3 -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last ~ 400 kernels before the end of the timed region: write compact trace
out = open("gpurun_out/trace/p3_compact.csv", "w")
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    out.write(f'{r["Kernel_Name"][:40]},{r.get("Stream_Id", r.get("Queue_Id"))},{int(r["Start_Timestamp"])-t0},{int(r["End_Timestamp"])-t0},{r.get("Workgroup_Size_X","")},{r.get("Grid_Size_X","")}\n')
print(len(rows), rows[0].keys())
PY
rm -rf gpurun_out/trace/p3
ls -la gpurun_out/trace
