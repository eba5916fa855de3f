import collections, csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
ro = [i for i, r in enumerate(rows) if "read_only_kernel" in r[2]]
# three marker calls, each a few launches close together: the last launch of each group
marks = [i for k, i in enumerate(ro) if k + 1 == len(ro) or rows[ro[k + 1]][0] - rows[i][1] > 1_000_000]
firsts = [i for k, i in enumerate(ro) if k == 0 or rows[i][0] - rows[ro[k - 1]][1] > 1_000_000]
print("marker groups", list(zip(firsts, marks)))
def seg(a, b, name):
    win = rows[a + 1:b]
    if not win:
        print(name, "empty"); return
    lo, hi = win[0][0], max(r[1] for r in win)
    span = (hi - lo) / 1e6
    busy = collections.defaultdict(float)
    tot = 0.0
    for s, e, k, q, st in win:
        busy[q] += (e - s) / 1e6
        tot += (e - s) / 1e6
    print(f"{name}: {len(win)} launches in {span:.2f} ms; kernel time {tot:.2f} ms = {tot / span:.2f} x the span; per queue busy:",
          {q: round(b / span, 2) for q, b in sorted(busy.items())})
    byk = collections.defaultdict(lambda: [0, 0.0])
    for s, e, k, q, st in win:
        byk[k][0] += 1; byk[k][1] += (e - s) / 1e3
    for k, (n, t) in sorted(byk.items(), key=lambda x: -x[1][1])[:8]:
        print(f"      {k:44s} {n:5d} launches, mean {t / n:8.1f} us")
seg(marks[0], firsts[1], "graph replays")
seg(marks[1], firsts[2], "stream-launched")
