"""Per-queue busy table of ONE rank's share of a W = 8 partition (scripts/r05_rank_share.py under rocprofv3
--kernel-trace): the trace is cut to the last 40 % of the window between the first and the last shard_unpack kernel
(the timed windows), then: batch period, kernel time / span (= kernels in flight on average), busy fraction per
hardware queue, the distribution of "how many kernels are in flight", and per kernel: launches, mean duration, ms per
batch.  usage: python scripts/r06_rank_share_trace.py <dir with *_kernel_trace.csv> [out.txt]"""
import collections
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:64],
                         "q%s" % r.get("Queue_Id", "?")))
    rows.sort()
    marks = [s for s, e, k, q in rows if "shard_unpack" in k]
    if not marks:
        print("no sharded kernels in the trace")
        return
    lo, hi = marks[0], marks[-1]
    lo = lo + int((hi - lo) * 0.6)
    win = [r for r in rows if r[0] >= lo and r[1] <= hi]
    span = (hi - lo) / 1e6
    nb = sum(1 for s, e, k, q in win if "shard_unpack" in k)
    out = [f"window {span:.2f} ms, {len(win)} launches, {nb} sharded batches: {span / max(nb, 1):.3f} ms per batch"]
    by_k = collections.defaultdict(lambda: [0, 0.0])
    by_q = collections.defaultdict(float)
    ev = []
    for s, e, k, q in win:
        by_k[k][0] += 1
        by_k[k][1] += (e - s) / 1e6
        by_q[q] += (e - s) / 1e6
        ev.append((s, 1))
        ev.append((e, -1))
    tot = sum(v[1] for v in by_k.values())
    out.append(f"kernel time {tot:.2f} ms = {tot / span:.2f} kernels in flight on average; per batch {tot / max(nb, 1):.3f} ms")
    for q, b in sorted(by_q.items()):
        out.append(f"  queue {q:6s} busy {b / span:5.2f} of the window")
    ev.sort()
    depth, last, hist = 0, lo, collections.defaultdict(float)
    for t, dlt in ev:
        hist[depth] += (t - last) / 1e6
        last = t
        depth += dlt
    out.append("kernels in flight: " + "  ".join(f"{k}: {v / span:.2f}" for k, v in sorted(hist.items())))
    out.append("kernel                                                            launches  mean us  ms/batch  share")
    for k, (n, b) in sorted(by_k.items(), key=lambda x: -x[1][1])[:40]:
        out.append(f"  {k:64s} {n:6d} {b / n * 1e3:8.1f} {b / max(nb, 1):9.4f} {b / tot:6.3f}")
    txt = "\n".join(out)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
