#!/usr/bin/env python3
"""Generate the golden vectors in this directory from the UNMODIFIED reference.

Runs only in the build container (needs /root/reference, Cython, a C++ compiler).
It copies the reference package to a temp dir, compiles its two Cython kernels
there with the flags of the reference's setup.py (setup.py:16-26,42-44) and
imports it; nothing of the reference is copied into this repository — the
fixtures are inputs and the outputs the reference produced for them.

Two compatibility shims are needed, neither touches reference sources:
  * Cython 3: compiler directive legacy_implicit_noexcept=True;
  * numpy 2: `numpy.core._methods` alias for fast_pq.py:15.

    python tests/golden/make_golden.py            # (re)write tests/golden/*.npz
"""
import os
import shutil
import subprocess
import sys
import tempfile
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

_BUILD = r'''
from setuptools import setup, Extension
from Cython.Build import cythonize
import numpy as np
args = ["-O3", "-march=native", "-ffast-math", "-Wno-unused-function", "-fprefetch-loop-arrays"]
exts = [
    Extension("tinyknn._fast_pq", ["tinyknn/_fast_pq.pyx"], extra_compile_args=args,
              language="c++", include_dirs=[np.get_include()]),
    Extension("tinyknn._fast_pq_avx", ["tinyknn/_fast_pq_256.pyx"],
              extra_compile_args=args + ["-mavx"], language="c++",
              include_dirs=[np.get_include()]),
]
setup(packages=["tinyknn"],
      ext_modules=cythonize(exts, compiler_directives={"legacy_implicit_noexcept": True}),
      script_args=["build_ext", "--inplace"])
'''


def import_reference(workdir=None):
    workdir = workdir or os.environ.get("TINYKNN_REF_BUILD") or os.path.join(
        tempfile.gettempdir(), "tinyknn_ref_build")
    pkg = os.path.join(workdir, "tinyknn")
    if not any(f.startswith("_fast_pq_avx") and f.endswith(".so")
               for f in (os.listdir(pkg) if os.path.isdir(pkg) else [])):
        os.makedirs(workdir, exist_ok=True)
        if os.path.isdir(pkg):
            shutil.rmtree(pkg)
        shutil.copytree(os.path.join(REF, "tinyknn"), pkg)
        with open(os.path.join(workdir, "build.py"), "w") as f:
            f.write(_BUILD)
        subprocess.check_call([sys.executable, "build.py"], cwd=workdir,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    import numpy
    sys.modules["numpy.core._methods"] = numpy._core._methods
    sys.path.insert(0, workdir)
    import tinyknn  # noqa
    return tinyknn


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    only = set(sys.argv[2:]) if len(sys.argv) > 2 and sys.argv[1] == "--only-ivf" else None
    warnings.filterwarnings("ignore")
    tk = import_reference()
    from tinyknn import _transform as rt
    from tinyknn._fast_pq import estimate_pq_sse, query_pq_sse, init_heap, insert, insert_is
    from tinyknn._fast_pq_avx import estimate_pq_avx, query_pq_avx
    from tinyknn.utils import knn_brute1, pad1
    assert tk.avx is True

    if only:
        return make_ivf(tk, only, query_pq_avx, knn_brute1)
    # ---- G1/G2: layout (mirrors tests/test_transform.py:71-101) -------------
    np.random.seed(10)
    out = {}
    for tag, (n, M) in {"a": (208, 14), "b": (64, 52)}.items():
        codes = np.random.randint(16, size=(n, M)).astype(np.uint8)
        tab = np.random.randint(256, size=(M, 16)).astype(np.uint8)
        out[f"codes_{tag}"] = codes
        out[f"packed_{tag}"] = rt.transform_data(codes)
        out[f"unpacked_{tag}"] = rt.unpack(out[f"packed_{tag}"]).astype(np.uint8)
        out[f"table_{tag}"] = tab
        out[f"ttable_{tag}"] = rt.transform_tables(tab)
    save("g1_layout.npz", **out)

    # ---- G3: estimate_pq (mirrors tests/test_pq.py:12-53) -------------------
    np.random.seed(10)
    out = {}
    cases = [(16, 4), (32, 8), (64, 52), (160, 32), (48, 6)]
    for ci, (n, M) in enumerate(cases):
        codes = np.random.randint(0, 16, size=(n, M), dtype=np.uint8)
        tab = np.random.randint(0, 256, size=(M, 16), dtype=np.uint8)
        if ci == 4:  # realistic range incl. negatives: exercises both rails less
            tab = np.random.randint(-4, 24, size=(M, 16)).astype(np.int8).view(np.uint8)
        d, t = rt.transform_data(codes), rt.transform_tables(tab)
        out[f"codes_{ci}"], out[f"table_{ci}"] = codes, tab
        for signed in (True, False):
            for name, fn in (("sse", estimate_pq_sse), ("avx", estimate_pq_avx)):
                o = np.zeros(n // 8, dtype=np.uint64)
                fn(d, t, o, signed)
                out[f"out_{ci}_{int(signed)}_{name}"] = o.view(np.uint8).copy()
    # the known-answer test of tests/test_transform.py:10-17
    codes = np.array([[1, 3, 7, 15]] + [[0, 0, 0, 0]] * 15, dtype=np.uint8)
    tab = np.array([list(range(16)) for _ in range(4)], dtype=np.uint8)
    o = np.zeros(2, dtype=np.uint64)
    estimate_pq_sse(rt.transform_data(codes), rt.transform_tables(tab), o, False)
    out["kat_codes"], out["kat_table"], out["kat_out"] = codes, tab, o.view(np.uint8).copy()
    save("g3_estimate.npz", **out)

    # ---- G4: query_pq heap arrays, layout included ---------------------------
    rng = np.random.default_rng(10)
    out = {}
    ci = 0
    meta = []
    for R in (1, 2, 3, 10, 21, 30, 111):
        for n in (1, 5, 17, 100, 599):
            M = int(rng.choice([4, 8, 32, 52]))
            signed = bool((ci % 3) != 2)
            use_labels = bool(ci % 2)
            npad = n + (-n) % 16
            codes = rng.integers(0, 16, size=(npad, M)).astype(np.uint8)
            if signed:  # heavy ties, both signs
                tab = rng.integers(-6, 12, size=(M, 16)).astype(np.int8).view(np.uint8)
            else:
                tab = rng.integers(0, 8, size=(M, 16)).astype(np.uint8)
            codes2 = rng.integers(0, 16, size=(npad, M)).astype(np.uint8)
            d1, d2, t = rt.transform_data(codes), rt.transform_data(codes2), rt.transform_tables(tab)
            labels1 = labels2 = None
            if use_labels:  # overlapping label sets => dedupe across lists; >2^32 too
                base = 10**12 if ci % 4 == 1 else 0
                labels1 = rng.integers(0, max(2, n), size=npad).astype(np.int64) + base
                labels2 = rng.integers(0, max(2, n), size=npad).astype(np.int64) + base
            for name, fn in (("sse", query_pq_sse), ("avx", query_pq_avx)):
                if name == "sse" and use_labels:
                    continue  # Cython-3 build of the SSE module truncates int64 labels
                idx = np.zeros(R, np.int64)
                val = np.zeros(R, np.int32)
                init_heap(idx, val, signed)
                # list 1, list 2, then list 1 again through ONE heap (ivf.py:137-150)
                for dd, ll in ((d1, labels1), (d2, labels2), (d1, labels1)):
                    if ll is None:
                        fn(dd, n, t, idx, val, signed)
                    else:
                        fn(dd, n, t, idx, val, signed, ll)
                out[f"idx_{ci}_{name}"], out[f"val_{ci}_{name}"] = idx, val
            out[f"codes1_{ci}"], out[f"codes2_{ci}"], out[f"table_{ci}"] = codes, codes2, tab
            if use_labels:
                out[f"labels1_{ci}"], out[f"labels2_{ci}"] = labels1, labels2
            meta.append((ci, R, n, M, int(signed), int(use_labels)))
            ci += 1
    out["meta"] = np.array(meta, dtype=np.int64)
    # heap micro-KATs (tests/test_heap.py:23-49) and a random insert trace
    idx = np.empty(3, np.int64); val = np.empty(3, np.int32); init_heap(idx, val, True)
    out["heap_init_idx"], out["heap_init_val"] = idx.copy(), val.copy()
    idx = np.empty(2, np.int64); val = np.empty(2, np.int32); init_heap(idx, val, True)
    insert(idx, val, 1, 10); insert(idx, val, 1, 10)
    out["heap_two_idx"], out["heap_two_val"] = idx.copy(), val.copy()
    ops = np.stack([rng.integers(0, 40, size=400), rng.integers(-30, 30, size=400)], axis=1).astype(np.int64)
    idx = np.empty(13, np.int64); val = np.empty(13, np.int32); init_heap(idx, val, True)
    idx2, val2 = idx.copy(), val.copy()
    tr_i, tr_v, tr_i2, tr_v2 = [], [], [], []
    for lab, v in ops:
        insert(idx, val, int(lab), int(v)); insert_is(idx2, val2, int(lab), int(v))
        tr_i.append(idx.copy()); tr_v.append(val.copy()); tr_i2.append(idx2.copy()); tr_v2.append(val2.copy())
    out["heap_ops"] = ops
    out["heap_trace_idx"], out["heap_trace_val"] = np.array(tr_i), np.array(tr_v)
    out["heap_is_trace_idx"], out["heap_is_trace_val"] = np.array(tr_i2), np.array(tr_v2)
    save("g4_query.npz", **out)

    # ---- G5/G7: distance tables + dequantised estimates ----------------------
    np.random.seed(10)
    out = {}
    meta = []
    for ci, (d, dpb, n) in enumerate([(100, 2, 600), (128, 2, 600), (20, 2, 300), (10, 1, 200),
                                      (24, 4, 300), (100, 1, 400), (11, 2, 100)]):
        X = np.random.randn(n, d).astype(np.float32)
        pq = tk.FastPQ(dpb)
        td = pq.fit_transform(X)
        qs = np.random.randn(8, d).astype(np.float32)
        qs[1] *= 0.05
        qs[2] *= 30
        out[f"centers_{ci}"] = pq.centers  # keeps its memory order through np.savez? stored C; flag below
        out[f"qs_{ci}"] = qs
        out[f"packed_{ci}"] = td.packed
        if pq.R is not None:
            out[f"R_{ci}"] = pq.R
        tabs, shifts, scales, est, uest = [], [], [], [], []
        utabs, ushifts, uscales, qpq = [], [], [], []
        for q in qs:
            dt = pq.distance_table(q)
            tabs.append(dt.tables); shifts.append(dt.mean); scales.append(dt.scale)
            qpq.append(dt.q)
            est.append(dt.estimate_distances(td, rescale=True))
            ut = pq.udistance_table(q)
            utabs.append(ut.tables); ushifts.append(ut.mean); uscales.append(ut.scale)
            uest.append(ut.estimate_distances(td).copy())
        out[f"tables_{ci}"] = np.array(tabs)
        out[f"shift_{ci}"] = np.array(shifts)
        out[f"scale_{ci}"] = np.array(scales)
        out[f"qpq_{ci}"] = np.array(qpq)
        out[f"est_rescaled_{ci}"] = np.array(est)
        out[f"utables_{ci}"] = np.array(utabs)
        out[f"ushift_{ci}"] = np.array(ushifts)
        out[f"uscale_{ci}"] = np.array(uscales)
        out[f"uest_{ci}"] = np.array(uest)
        meta.append((ci, d, dpb, n, td.size, int(pq.R is not None),
                     int(not pq.centers.flags.c_contiguous)))
        out[f"sqrt_n_blocks_{ci}"] = np.float64(pq.sqrt_n_blocks)
    out["meta"] = np.array(meta, dtype=np.int64)
    save("g5_tables.npz", **out)

    make_ivf(tk, None, query_pq_avx, knn_brute1)


def make_ivf(tk, only, query_pq_avx, knn_brute1):
    # ---- G6: IVF end to end ---------------------------------------------------
    for tag, (n, d, metric, ncl, bprobes, nq, dtype) in {
        "eu20": (2000, 20, "euclidean", 44, 2, 24, np.float32),
        "an20": (2000, 20, "angular", 44, 1, 24, np.float32),
        "an100": (2000, 100, "angular", 44, 1, 24, np.float32),
        "an100b2": (1000, 100, "angular", 31, 2, 16, np.float32),
        "eu128": (1200, 128, "euclidean", 34, 1, 16, np.float32),
        "eu20f64": (1500, 20, "euclidean", 38, 1, 16, np.float64),   # float64 X: float64 rescoring
    }.items():
        if only and tag not in only:
            continue
        np.random.seed(10)
        # clustered data so that int8 sums use both rails
        cent = np.random.randn(30, d)
        X = (cent[np.random.randint(30, size=n)] + 0.7 * np.random.randn(n, d)).astype(dtype)
        qs = (cent[np.random.randint(30, size=nq)] + 0.7 * np.random.randn(nq, d)).astype(np.float32)
        ivf = tk.IVF(metric, ncl, tk.FastPQ(2))
        ivf.fit(X).build(X, n_probes=bprobes)
        nl = len(ivf.active_centers)
        out = dict(
            qs=qs, metric=np.array(metric), build_probes=np.int64(bprobes),
            pq_centers=ivf.pq.centers, sqrt_n_blocks=np.float64(ivf.pq.sqrt_n_blocks),
            active_centers=ivf.active_centers,
            center_size=np.int64(ivf.pq_transformed_centers.size),
            center_codes=ivf.pq_transformed_centers.packed,
            data=ivf.data,
            list_sizes=np.array([ivf.pq_transformed_points[i].size for i in range(nl)], dtype=np.int64),
            list_codes=np.concatenate([ivf.pq_transformed_points[i].packed for i in range(nl)]),
            ids=np.concatenate([np.asarray(ivf.ids[i], dtype=np.int64) for i in range(nl)]),
        )
        if ivf.pq.R is not None:
            out["R"] = ivf.pq.R
        probes_list = (1, 2, 5, 10)
        k = 10
        qn_all, qpq_all = [], []
        for pi, n_probes in enumerate(probes_list):
            res, probes, hidx, hval, tabs = [], [], [], [], []
            for q in qs:
                ids_ref = ivf.query(q.copy(), k, n_probes=n_probes)
                # replay ivf.py:125-150 with the reference's own pieces to expose
                # the intermediate state
                qn = np.ascontiguousarray(q.copy(), dtype=np.float32)
                if metric == "angular":
                    qn /= np.linalg.norm(qn)
                dtable = ivf.pq.distance_table(qn)
                top = dtable.top(ivf.pq_transformed_centers, ivf.active_centers, k=n_probes)
                pass_1 = (n_probes + 1) * k + 1
                indices = np.full(pass_1, -1, dtype=np.int64)
                values = np.full(pass_1, 127, dtype=np.int32)
                for cl in top:
                    true_n, tdata = ivf.pq_transformed_points[cl]
                    query_pq_avx(tdata, true_n, dtable.tables, indices, values, True, labels=ivf.ids[cl])
                heap_i, heap_v = indices.copy(), values.copy()
                if -1 in indices:
                    indices = indices[indices != -1]
                if len(indices) > k:
                    indices = indices[knn_brute1(qn, ivf.data[indices], k)]
                assert np.array_equal(indices, ids_ref)
                r = np.full(k, -1, np.int64); r[:len(ids_ref)] = ids_ref
                res.append(r); probes.append(np.asarray(top, np.int64)); hidx.append(heap_i); hval.append(heap_v)
                tabs.append(dtable.tables)
                if pi == 0:
                    qn_all.append(qn); qpq_all.append(dtable.q)
            out[f"ids_p{n_probes}"] = np.array(res)
            out[f"probes_p{n_probes}"] = np.array(probes)
            out[f"heap_idx_p{n_probes}"] = np.array(hidx)
            out[f"heap_val_p{n_probes}"] = np.array(hval)
            if pi == 0:
                out["tables"] = np.array(tabs)
        out["qn"] = np.array(qn_all)
        out["qpq"] = np.array(qpq_all)
        out["probes_list"] = np.array(probes_list, dtype=np.int64)
        save(f"g6_ivf_{tag}.npz", **out)


if __name__ == "__main__":
    main()
