"""Raw queries (host) in -> ids (host) out through a streaming session, timed alone.
usage: python scripts/raw_stream_probe.py [--steps 200] [--slots 8] [--prepared]
env TINYKNN_STREAM_COPY=0|1 selects copy engine / zero-copy kernels (front.hip)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--slots", type=int, default=8)
    ap.add_argument("--nq", type=int, default=10000)
    ap.add_argument("--n-probes", type=int, default=10)
    ap.add_argument("--pipeline", type=int, default=2)
    ap.add_argument("--prepared", action="store_true", help="skip the host preparation (submit_prepared)")
    a = ap.parse_args()
    import torch
    import bench
    from tinyknn_amd import _front
    args = argparse.Namespace(n=1183514, d=100, n_clusters=1087, seed=10, build_probes=1, metric="angular",
                              data="glove-like", fit_sample=100000, cache_dir=os.environ.get("TMPDIR", "/tmp"))
    torch.cuda.set_device(0)
    ivf, cent = bench.build_index(args, torch.device("cuda", 0))
    dev = ivf.device_index()
    dev.set_pipeline(a.pipeline)
    qs = bench.synth_queries(cent, a.nq, 110, kind="glove-like")
    assert _front.bind()
    st = dev.stream(a.nq, 10, a.n_probes, slots=a.slots)
    outs = [np.full((a.nq, 10), -1, dtype=np.int64) for _ in range(a.slots)]
    qn, qp = ivf._prepare(qs.copy())
    sub = (lambda o: st.submit_prepared(qn, qp, o)) if a.prepared else (lambda o: st.submit(qs, o))
    for i in range(2 * a.slots):
        sub(outs[i % a.slots])
    st.drain()
    t0 = time.perf_counter()
    for i in range(a.steps):
        sub(outs[i % a.slots])
    st.drain()
    el = time.perf_counter() - t0
    want = dev.query_batch(qn, qp, 10, a.n_probes)
    same = int(min((o == want).all(axis=1).sum() for o in outs))
    st.close()
    print(json.dumps({"copy_mode": os.environ.get("TINYKNN_STREAM_COPY", "1"), "prepared": a.prepared,
                      "slots": a.slots, "pipeline": a.pipeline, "ms_per_step": el / a.steps * 1e3,
                      "queries_per_s": a.nq * a.steps / el, "rows_identical": same, "rows": a.nq}))


if __name__ == "__main__":
    main()
