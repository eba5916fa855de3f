# kernel timeline of the pipelined default run (start/end per kernel, per queue): gaps on the scan chain
R=$PWD; O=$R/gpurun_out/r03_trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/bench.py --traffic none --shard none --pipeline 2 --profile-only --steps 20 --windows 3 --warmup-seconds 0.3 "$@" > $O/bench.json 2> $O/kt.err
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" $O <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), "kernel records; columns:", list(rows[0].keys()))
# keep the last 4000 records (steady state), trimmed columns
keep = rows[-6000:]
with open(sys.argv[2] + "/trace_tail.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["name", "queue", "stream", "start", "end"])
    for r in keep:
        w.writerow([r["Kernel_Name"][:48], r.get("Queue_Id", ""), r.get("Stream_Id", ""), r["Start_Timestamp"], r["End_Timestamp"]])
PY
rm -rf $O/kt
