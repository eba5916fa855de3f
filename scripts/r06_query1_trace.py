"""One ivf.query(q) per call under rocprofv3 --kernel-trace --memory-copy-trace: what a call consists of on the device
(kernels, copies, fills in start order with durations and the gaps between them), averaged over the last calls.
usage: rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python3 scripts/r06_query1_trace.py run
       python3 scripts/r06_query1_trace.py report DIR [out.txt]"""
import sys, os, glob, csv, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import argparse
    import numpy as np, torch
    import bench
    a = argparse.ArgumentParser().parse_args([])
    a.n, a.d, a.n_clusters, a.seed, a.build_probes, a.metric, a.data, a.fit_sample = 1183514, 100, 1087, 10, 1, "angular", "glove-like", 100000
    a.cache_dir, a.data_file, a.nq, a.k = os.environ.get("TMPDIR", "/tmp"), None, 10000, 10
    ivf, cent = bench.build_index(a, torch.device("cuda:0"))
    qs = bench.synth_queries(cent, 300, 12345, kind=a.data)
    for i in range(300):
        ivf.query(qs[i].copy(), k=10, n_probes=10)
    torch.cuda.synchronize()


def report(d, out=None):
    ev = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:56]))
    for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?")))
    ev.sort()
    # a call ends with the last rescoring kernel: cut the stream of events there, keep the last 100 calls
    calls, cur = [], []
    for e in ev:
        cur.append(e)
        if "rescore_staged_kernel" in e[2] or "rescore_kernel" in e[2]:
            pass
    # split on the table build (first kernel of a call)
    for e in ev:
        if "build_tables_kernel" in e[2] and cur:
            calls.append(cur)
            cur = []
        cur.append(e)
    calls = [c for c in calls[-120:-5] if any("build_tables" in x[2] for x in c)]
    shape = collections.Counter(tuple(x[2] for x in c) for c in calls).most_common(1)[0][0]
    same = [c for c in calls if tuple(x[2] for x in c) == shape]
    lines = [f"{len(same)} calls of the same shape ({len(shape)} device operations per call)",
             "op                                                        start us   dur us   gap before us"]
    n = len(same)
    tot_d = tot_g = 0
    for i, name in enumerate(shape):
        st = sum(c[i][0] - c[0][0] for c in same) / n / 1e3
        du = sum(c[i][1] - c[i][0] for c in same) / n / 1e3
        gp = sum((c[i][0] - c[i - 1][1]) if i else 0 for c in same) / n / 1e3
        tot_d += du
        tot_g += gp
        lines.append(f"  {name:56s} {st:8.1f} {du:8.1f} {gp:8.1f}")
    span = sum(c[-1][1] - c[0][0] for c in same) / n / 1e3
    period = (same[-1][0][0] - same[0][0][0]) / max(len(same) - 1, 1) / 1e3
    lines.append(f"device span of a call {span:.1f} us = busy {tot_d:.1f} + gaps {tot_g:.1f}; call period (profiled) {period:.1f} us")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
