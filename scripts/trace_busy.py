"""Per-stream busy fractions and the scan chain's gaps from a rocprofv3 kernel trace (+ memory
copy trace if present).  usage: python scripts/trace_busy.py <dir with *_kernel_trace.csv> [skip_frac]"""
import collections
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in kt:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48],
                         "q%s/s%s" % (r.get("Queue_Id", "?"), r.get("Stream_Id", "?"))))
    for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", ""),
                         "copy/s%s" % r.get("Stream_Id", "?")))
    rows.sort()
    t0, t1 = rows[0][0], rows[-1][1]
    lo = t0 + int((t1 - t0) * skip)
    hi = t1 - int((t1 - t0) * 0.1)
    win = [r for r in rows if r[0] >= lo and r[1] <= hi]
    span = (hi - lo) / 1e6
    print(f"window {span:.2f} ms, {len(win)} records")
    by_q = collections.defaultdict(float)
    by_k = collections.defaultdict(lambda: [0, 0.0])
    for s, e, k, q in win:
        by_q[q] += (e - s) / 1e6
        by_k[(q, k)][0] += 1
        by_k[(q, k)][1] += (e - s) / 1e6
    for q, b in sorted(by_q.items()):
        print(f"  {q:14s} busy {b:8.2f} ms = {b / span:5.2f} of the window")
    print("per kernel (queue, name, launches, mean us, share of window):")
    for (q, k), (n, b) in sorted(by_k.items(), key=lambda x: -x[1][1])[:24]:
        print(f"  {q:14s} {k:48s} {n:5d} {b / n * 1e3:8.1f} {b / span:6.3f}")
    sc = [(s, e) for s, e, k, q in win if "scan_units2" in k or k.startswith("void scan_units")]
    if len(sc) > 2:
        gaps = [(sc[i + 1][0] - sc[i][1]) / 1e3 for i in range(len(sc) - 1)]
        durs = [(e - s) / 1e3 for s, e in sc]
        print(f"scan launches {len(sc)}: mean {sum(durs) / len(durs):.0f} us, mean gap {sum(gaps) / len(gaps):.0f} us, "
              f"period {(sc[-1][0] - sc[0][0]) / (len(sc) - 1) / 1e3:.0f} us")


if __name__ == "__main__":
    main()
