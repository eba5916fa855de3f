R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
TINYKNN_SHARD_ROLES=1 timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt9 -- python3 $R/scripts/r05_rank_share.py --steps 20 --depth 4 > $O/run9_rank_share.json 2> $O/run9_rank_share.err
echo "rc=$?"; cd $R
python3 - <<'PY'
import csv, glob, json
rows = []
for f in glob.glob("gpurun_out/r05/kt9/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
marks = [i for i, r in enumerate(rows) if "shard_unpack" in r[2]]
i0 = marks[-9]
t0 = rows[i0][0]
last = {}
out = []
for s, e, k, q, st in rows[i0:marks[-1] + 40]:
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    if e - s > 15000 or "at::" not in k:
        out.append(f"{(s - t0) / 1e3:9.1f} us  q{q:>2s} s{st:>3s}  {(e - s) / 1e3:7.1f} us  gap {gap:7.1f}  {k}")
open("gpurun_out/r05/run9_timeline.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:150]))
j = json.loads([l for l in open("gpurun_out/r05/run9_rank_share.json") if l.startswith("{")][-1])
print({k: v for k, v in j.items() if k in ("ms_per_step", "host_enqueue_ms_per_step", "windows_ms")})
PY
rm -rf $O/kt9
