R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt6 -- python3 $R/scripts/r05_rank_share.py --steps 20 --depth 4 > $O/run6_rank_share.json 2> $O/run6_rank_share.err
echo "rc=$?"; cd $R
python3 scripts/r05_shard_trace.py $O/kt6 $O/run6_rank_share_trace.txt | head -50
python3 - <<'PY'
import csv, glob, json
rows = []
for f in glob.glob("gpurun_out/r05/kt6/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
marks = [i for i, r in enumerate(rows) if "shard_unpack" in r[2]]
# the last 6 batches: every launch with queue, start offset, duration, gap to the previous launch of its queue
i0 = marks[-7]
t0 = rows[i0][0]
last = {}
out = []
for s, e, k, q, st in rows[i0:marks[-1] + 40]:
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    out.append(f"{(s - t0) / 1e3:9.1f} us  q{q:>2s} s{st:>3s}  {(e - s) / 1e3:7.1f} us  gap {gap:7.1f}  {k}")
open("gpurun_out/r05/run6_timeline.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:130]))
j = json.loads([l for l in open("gpurun_out/r05/run6_rank_share.json") if l.startswith("{")][-1])
print({k: v for k, v in j.items() if k in ("ms_per_step", "host_enqueue_ms_per_step", "windows_ms")})
PY
rm -rf $O/kt6
