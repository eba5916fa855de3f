"""Per-kernel time inside the LIST-SHARDED leg of a profiled bench.py run: the rocprofv3 kernel trace is cut
to the window between the first and the last sharded-only kernel (shard_lens / shard_unpack), so the headline
region's launches do not mix in.  usage: python scripts/r05_shard_trace.py <dir with *_kernel_trace.csv> [out.txt]"""
import collections
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60],
                         "q%s" % r.get("Queue_Id", "?")))
    rows.sort()
    marks = [s for s, e, k, q in rows if "shard_unpack" in k or "shard_lens" in k]
    if not marks:
        print("no sharded kernels in the trace")
        return
    lo, hi = marks[0], marks[-1]
    # the timed windows are the last ~60 % of the leg (settling + warm-up come first)
    lo = lo + int((hi - lo) * 0.4)
    win = [r for r in rows if r[0] >= lo and r[1] <= hi]
    span = (hi - lo) / 1e6
    nb = sum(1 for s, e, k, q in win if "shard_unpack" in k)
    out = [f"window {span:.1f} ms, {len(win)} launches, {nb} sharded batches ({span / max(nb, 1):.3f} ms per batch)"]
    by_k = collections.defaultdict(lambda: [0, 0.0])
    by_q = collections.defaultdict(float)
    for s, e, k, q in win:
        by_k[k][0] += 1
        by_k[k][1] += (e - s) / 1e6
        by_q[q] += (e - s) / 1e6
    tot = sum(v[1] for v in by_k.values())
    out.append(f"kernel time {tot:.1f} ms = {tot / span:.2f} x the window; per batch {tot / max(nb, 1):.3f} ms")
    for q, b in sorted(by_q.items()):
        out.append(f"  {q:8s} busy {b / span:5.2f} of the window")
    out.append("kernel                                                        launches  mean us  ms/batch  share")
    for k, (n, b) in sorted(by_k.items(), key=lambda x: -x[1][1])[:40]:
        out.append(f"  {k:60s} {n:6d} {b / n * 1e3:8.1f} {b / max(nb, 1):9.4f} {b / tot:6.3f}")
    txt = "\n".join(out)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
