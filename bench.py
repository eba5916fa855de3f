#!/usr/bin/env python3
"""bench.py — queries/sec of the IVF + 4-bit-PQ hot path on MI355X.

Workload (BASELINE.json configs[1], GloVe-100 stand-in — no network, so the data is
synthetic with GloVe-100's shape): N = 1 183 514 x 100 float32 drawn
from 300 Gaussian clusters (sigma 0.7, seed 10), angular metric, IVF with
n_clusters = 1087 = int(sqrt(N)), build_probes = 1, FastPQ(dims_per_block=2)
(M = 52 blocks, 26 B/code), k = 10, n_probes = 10, one step = one batch of 10 000
queries through the whole device pipeline (tables, coarse stage, list scan, exact
heap replay, exact rescoring).  Queries are resident in HBM, already normalised as
the reference's host code does (ivf.py:125-127), when the timed region starts.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: `value` is the LIST-SHARDED index (north_star's split): inverted
lists partitioned by cluster id over the ranks, ONE shared batch of --nq queries per
step ("strong": the work per step does not grow with N), per batch the coarse stage
of a rank's home queries, an all-gather of the probe lists, the scan of the owned
(query, list) segments, one RCCL all-to-all of int8 distance bytes, the exact replay
on the home rank and the all-gather of the ids; its ids are checked against the
unsharded pipeline's.  The replica rate (every rank holds the whole index and answers
its own batch, no data-path collective, "weak") is measured first and reported under
"replica"; it is what `value` falls back to, with "list_sharded.error" set and a
non-zero exit code, if the sharded leg fails (`--shard lists` forces the leg at N = 1,
`--shard none` skips it).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


def log(*a):
    print(*a, file=sys.stderr, flush=True)


_T0 = [time.time()]


def lap(what):
    """stderr: seconds since the previous lap (where a run's wall time goes, leg by leg)."""
    now = time.time()
    log(f"[bench] {what}: {now - _T0[0]:.1f}s")
    _T0[0] = now

COMPACT_LIMIT = 4096     # bytes of the FINAL stdout line (the driver keeps a bounded tail of stdout)


def _r(x, nd=6):
    """Round floats for the compact line (the detail keeps the full figures)."""
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{nd}g}")
    return x


def _pick(d, *keys, nd=6):
    if not isinstance(d, dict):
        return None
    return {k: _r(d.get(k), nd) for k in keys if d.get(k) is not None}


def compact_line(line, detail_file=None):
    """The FINAL stdout line: the driver contract's keys + `roofline` + `cpu_baseline` + `parity_vs_oracle` + a few
    scalars, <= COMPACT_LIMIT bytes whatever the legs produced.  Everything else (sweep points, windows of the
    sharded legs, hipGraph, isolated stages, the notes) goes into the DETAIL record: an earlier stdout line
    prefixed `# bench_detail ` and `gpurun_out/bench/bench_detail.json`.  (Round 5's single line had grown to 20.9 KB
    and the driver's bounded tail lost its head: metric, value, config, roofline.)"""
    cfg = line.get("config") or {}
    roof = line.get("roofline") if isinstance(line.get("roofline"), dict) else {}
    out = {
        "metric": str(line.get("metric"))[:200], "value": _r(line.get("value"), 8), "unit": line.get("unit"),
        "n_gpus": line.get("n_gpus"), "steps": line.get("steps"), "warmup": line.get("warmup"),
        "ms_per_step": _r(line.get("ms_per_step"), 8), "higher_is_better": True,
        "scaling": line.get("scaling"), "vs_baseline": line.get("vs_baseline"),
        "dtype": line.get("dtype"), "data": line.get("data"),
        "config": {"workload": str(cfg.get("workload"))[:260], "k": cfg.get("k"), "n_probes": cfg.get("n_probes"),
                   "recall10@10": _r(cfg.get("recall10@10"), 4),
                   "queries_per_step_per_gpu": cfg.get("queries_per_step_per_gpu"),
                   "parallelism": str(cfg.get("parallelism"))[:200], "batches_in_flight": cfg.get("batches_in_flight")},
    }
    if cfg.get("queries_per_step_total") is not None:
        out["config"]["queries_per_step_total"] = cfg["queries_per_step_total"]
    rf = _pick(roof, "bound", "kernel", "achieved", "peak", "unit", "frac", "kernel_ms", "kernel_ms_rocprofv3",
               "traffic", "hbm_frac_measured", "algorithmic_bytes_per_launch", "launch_covers", "mfma_floor_ms")
    if rf is not None:
        rf.setdefault("traffic", None)
        if "kernel" in rf:
            rf["kernel"] = str(rf["kernel"])[:80]
        if isinstance(roof.get("plain_scan"), dict) and roof["plain_scan"].get("tile_fill") is not None:
            rf["tile_fill"] = _r(roof["plain_scan"]["tile_fill"], 4)
        if isinstance(roof.get("exact_launch"), dict):
            rf["exact_launch"] = _pick(roof["exact_launch"], "bound", "frac", "kernel_ms_rocprofv3", nd=4)
        for key in ("replay", "rescore"):
            if isinstance(roof.get(key), dict):
                rf[key] = _pick(roof[key], "frac", "frac_timed_region", "kernel_ms_isolated", nd=4)
        ks = roof.get("kernel_stats_rocprofv3")
        if isinstance(ks, dict):      # share of kernel time of the biggest consumers (name up to the template list)
            rf["kernel_time_pct"] = {re.sub(r"^void ", "", n).split("<")[0].split("(")[0][:32]: _r(e.get("pct"), 3)
                                     for n, e in list(ks.items())[:5] if isinstance(e, dict)}
        rf["kernel_stats_csv"] = roof.get("kernel_stats_csv")
    out["roofline"] = rf
    cpu = line.get("cpu_baseline")
    if isinstance(cpu, dict):
        c = _pick(cpu, "value", "unit", "cores", "kind")
        c["sample"] = str(cpu.get("sample"))[:140]
        if isinstance(cpu.get("python_loop"), dict):
            c["python_loop_value"] = _r(cpu["python_loop"].get("value"))
        out["cpu_baseline"] = c
    else:
        out["cpu_baseline"] = cpu
    out["parity_vs_oracle"] = line.get("parity_vs_oracle")
    # scalars
    for k, v in line.items():
        if isinstance(v, (int, float)) and not isinstance(v, bool) and k not in out and (
                k.startswith("rank_share_W") or k.startswith("roofline_") or k.startswith("list_sharded_")
                or k.startswith("raw_in_ids_out") or k.startswith("query1_")):
            out[k] = _r(v)
    hs = line.get("roofline_hbm_scale")
    if isinstance(hs, dict) and hs.get("frac") is not None:
        # the exact code scan where it streams HBM (1 GiB of codes, one query: every byte fetched once): GB/s and the fraction
        # of the 8 TB/s peak (north_star's ">= 60 % on the code-scan kernel" reads on this shape)
        out["scan_hbm_stream"] = {"GBps": _r(hs.get("GBps"), 5), "frac": _r(hs.get("frac"), 4),
                                  "frac_of_read_only_kernel": _r(hs.get("frac_of_measured_read_only_kernel"), 4)}
    g = line.get("hipgraph")
    if isinstance(g, dict) and g.get("queries_per_s") and line.get("value"):
        out["hipgraph_queries_per_s"] = _r(g["queries_per_s"])
        out["hipgraph_ratio"] = _r(g["queries_per_s"] / line["value"], 4)
        out["hipgraph_identical_to_stream_launch"] = g.get("identical_to_stream_launch")
    sw = line.get("sweep")
    if isinstance(sw, dict) and isinstance(sw.get("points"), list):
        pts = {}
        for pt in sw["points"]:
            c = str(pt.get("config", ""))
            tag = ("np%d" % pt.get("n_probes", 0)) if c.startswith("headline") else (
                "b2" if "n_probes=2" in c else "sift" if "sift" in c else c[:12])
            par = pt.get("parity_vs_oracle") or {}
            pts[tag] = [_r(pt.get("queries_per_s"), 4), par.get("identical_rows"), par.get("queries_checked")]
            if tag == "b2":
                out["sweep_b2_queries_per_s"] = _r(pt.get("queries_per_s"))
        out["sweep_points"] = pts      # tag: [queries/s, rows identical to the oracle, rows checked]
    elif isinstance(sw, dict) and sw.get("error"):
        out["sweep_error"] = str(sw["error"])[:120]
    ls = line.get("list_sharded")
    if isinstance(ls, dict):
        out["list_sharded"] = (_pick(ls, "queries_per_s", "ms_per_step", "identical_rows_vs_replica", "rows",
                                     "batches_in_flight", "steps_coalesced_per_exchange")
                               if "error" not in ls else {"error": str(ls["error"])[:160]})
    if isinstance(line.get("replica"), dict):
        out["replica"] = _pick(line["replica"], "queries_per_s", "ms_per_step")
    q1 = line.get("query1")
    if isinstance(q1, dict):
        out["query1"] = _pick(q1, "ms_per_query", "queries_per_s", "identical_rows", "rows", "cpu_oracle_ms_per_query", nd=4)
    out["detail"] = detail_file
    # hard guard: drop optional keys, least important first, until the line fits
    for drop in ("sweep_points", "query1", "scan_hbm_stream", "replica", "list_sharded", "hipgraph_identical_to_stream_launch",
                 "hipgraph_queries_per_s", "detail"):
        if len(json.dumps(out, allow_nan=False)) <= COMPACT_LIMIT:
            break
        out.pop(drop, None)
    if len(json.dumps(out, allow_nan=False)) > COMPACT_LIMIT and isinstance(out.get("roofline"), dict):
        for drop in ("kernel_time_pct", "exact_launch", "kernel_stats_csv", "launch_covers"):
            out["roofline"].pop(drop, None)
    return out


def _json_safe(x):
    if isinstance(x, float) and (x != x or x in (float("inf"), float("-inf"))):
        return None
    if isinstance(x, dict):
        return {str(k): _json_safe(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_json_safe(v) for v in x]
    if isinstance(x, np.generic):
        return _json_safe(x.item())
    return x


def emit(line, detail_file=None):
    """Detail first (file + a prefixed stdout line), then the compact line as the LAST line of stdout."""
    full = _json_safe(line)
    path = detail_file or os.path.join(ROOT, "gpurun_out", "bench", "bench_detail.json")
    shown = None
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f)
        shown = os.path.relpath(path, ROOT)
    except OSError as e:
        log("bench detail file not written:", e)
    print("# bench_detail " + json.dumps(full), flush=True)
    print(json.dumps(compact_line(full, shown), allow_nan=False), flush=True)



def synth(n, nq, d, seed, n_centres=300, sigma=0.7, kind="glove-like"):
    rng = np.random.RandomState(seed)
    cent = rng.randn(n_centres, d)
    X = np.empty((n, d), dtype=np.float32)
    step = 200000
    for i in range(0, n, step):
        m = min(step, n - i)
        if kind == "sift-like":      # SURVEY §8d C3 stand-in
            X[i:i + m] = np.clip(np.abs(rng.randn(m, d)) * 40, 0, 218).round()
        elif kind == "sift-clustered":   # the same value range with structure: 300 clusters, so that recall means something
            X[i:i + m] = np.clip(np.abs(cent[rng.randint(n_centres, size=m)]) * 40 + 12.0 * rng.randn(m, d), 0, 218).round()
        else:
            X[i:i + m] = cent[rng.randint(n_centres, size=m)] + sigma * rng.randn(m, d)
    return X, cent


def save_atomic(path, write):
    """write(file object) into `path` by way of a private name + rename: ranks and child runs that share the
    cache directory never read a half-written file."""
    tmp = "%s.%d.tmp" % (path, os.getpid())
    try:
        with open(tmp, "wb") as f:
            write(f)
        os.replace(tmp, path)
    except OSError as e:
        log(f"[bench] could not cache {os.path.basename(path)}: {e}")
        try:
            os.unlink(tmp)
        except OSError:
            pass


def synth_cached(args):
    """synth() of the run's data set, kept as a .npy beside the index cache: the profiler child runs and the sweep's
    second build over the same rows load it instead of drawing 118M normals again (7 s each)."""
    path = os.path.join(args.cache_dir, f"tinyknn_bench_rows_n{args.n}_d{args.d}_s{args.seed}_{args.data}.npy")
    cent = np.random.RandomState(args.seed).randn(300, args.d)      # synth()'s first draw
    if os.path.exists(path):
        try:
            X = np.load(path)
            if X.shape == (args.n, args.d) and X.dtype == np.float32:
                return X, cent
        except Exception:      # noqa: BLE001 - a damaged cache file is regenerated, whatever numpy says about it
            pass
    X, cent = synth(args.n, 0, args.d, args.seed, kind=args.data)
    save_atomic(path, lambda f: np.save(f, X))
    return X, cent


def synth_queries(cent, nq, seed, sigma=0.7, kind="glove-like"):
    rng = np.random.RandomState(seed)
    d = cent.shape[1]
    if kind == "sift-like":
        return np.clip(np.abs(rng.randn(nq, d)) * 40, 0, 218).round().astype(np.float32)
    if kind == "sift-clustered":
        return np.clip(np.abs(cent[rng.randint(len(cent), size=nq)]) * 40 + 12.0 * rng.randn(nq, d), 0, 218).round().astype(np.float32)
    return (cent[rng.randint(len(cent), size=nq)] + sigma * rng.randn(nq, d)).astype(np.float32)


def quick_kmeans(X, k, iters, seed, device):
    """Lloyd iterations for the coarse centres (offline set-up, not the measured
    path; the reference uses sklearn KMeans here, ivf.py:31-45)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    Xt = torch.from_numpy(X).to(device)
    C = Xt[torch.randperm(len(X), generator=g)[:k].to(device)].clone()
    for _ in range(iters):
        assign = torch.empty(len(X), dtype=torch.long, device=device)
        for i in range(0, len(X), 65536):
            xb = Xt[i:i + 65536]
            dist = (xb * xb).sum(1, keepdim=True) - 2 * xb @ C.T + (C * C).sum(1)[None]
            assign[i:i + 65536] = dist.argmin(1)
        sums = torch.zeros_like(C).index_add_(0, assign, Xt)
        cnt = torch.bincount(assign, minlength=k)
        C = sums / cnt.clamp(min=1).unsqueeze(1)
        empty = (cnt == 0).nonzero().flatten()
        if len(empty):  # re-seed empty clusters on data points so that every list is active
            C[empty] = Xt[torch.randperm(len(X), generator=g)[:len(empty)].to(device)]
    return C.cpu().numpy().astype(np.float64)


def load_real(args):
    """examples/bench.py:67-70: load the .npy, np.random.seed(10), shuffle, the last n_queries rows
    are the queries (unless --query-file holds them).  GloVe-100 / SIFT-1M run unchanged when the
    files are provided (there is no network here to fetch them)."""
    data = np.load(args.data_file)
    np.random.seed(10)
    np.random.shuffle(data)
    data = np.ascontiguousarray(data, dtype=np.float32)
    if args.query_file:
        return data, np.ascontiguousarray(np.load(args.query_file), dtype=np.float32)
    return data[:-args.nq], data[-args.nq:]


def build_index(args, device, X=None):
    """Fit + build with the product's host code; cached on local disk because the
    driver runs N = 1, 2, 4, 8 back to back on one box."""
    from tinyknn_amd import IVF, FastPQ
    from tinyknn_amd.fast_pq import TransformedData
    src = args.data if X is None else os.path.basename(args.data_file)
    tag = f"n{args.n}_d{args.d}_c{args.n_clusters}_s{args.seed}_b{args.build_probes}_{args.metric}_{src}_c32"
    cache = os.path.join(args.cache_dir, f"tinyknn_bench_{tag}.npz")
    cent = None
    if X is None:
        X, cent = synth_cached(args)
    ang = args.metric == "angular"
    ivf = IVF(args.metric, args.n_clusters, FastPQ(2))
    if os.path.exists(cache):
        try:
            z = np.load(cache)
            ivf.pq.centers = z["pq_centers"]
            ivf.pq.sqrt_n_blocks = float(z["sqrt_n_blocks"])
            ivf.pq.R = z["R"] if "R" in z else None
            ivf.active_centers = z["active_centers"]
            ivf.pq_transformed_centers = TransformedData(int(z["center_size"]), z["center_codes"])
            sizes, codes, ids = z["list_sizes"], z["list_codes"], z["ids"]      # (an NpzFile reads the member at EVERY z[...])
            coff = np.concatenate([[0], np.cumsum((sizes + 15) // 16)])
            ioff = np.concatenate([[0], np.cumsum(sizes)])
            ivf.pq_transformed_points = [TransformedData(int(sizes[i]), codes[coff[i]:coff[i + 1]])
                                         for i in range(len(sizes))]
            ivf.ids = [ids[ioff[i]:ioff[i + 1]] for i in range(len(sizes))]
            ivf.data = X / np.linalg.norm(X, axis=1, keepdims=True) if ang else X
            log(f"[bench] index loaded from {cache}")
            return ivf, cent
        except Exception as e:      # noqa: BLE001 - a damaged cache file: fit and build again
            log(f"[bench] cache {cache} not usable ({e!r}): building")
            ivf = IVF(args.metric, args.n_clusters, FastPQ(2))
    t0 = time.time()
    rng = np.random.RandomState(args.seed + 1)
    sample = X[rng.choice(len(X), min(len(X), args.fit_sample), replace=False)]
    if ang:
        sample = sample / np.linalg.norm(sample, axis=1, keepdims=True)
    C = quick_kmeans(sample, args.n_clusters, 8, args.seed, device)
    C = C.astype(np.float32)          # what sklearn's KMeans returns for float32 data (ivf.py:31-45)
    ivf.all_centers = C / np.linalg.norm(C, axis=1, keepdims=True) if ang else C   # ivf.py:36-45
    ivf.pq.fit(sample[:min(len(sample), 30000)])
    log(f"[bench] fit done in {time.time() - t0:.1f}s")
    ivf.build(X, n_probes=args.build_probes, device=True)   # build.hip: same lists and codes as numpy
    log(f"[bench] build done in {time.time() - t0:.1f}s")
    L = len(ivf.active_centers)
    extra = {} if ivf.pq.R is None else {"R": ivf.pq.R}
    save_atomic(cache, lambda f: np.savez(
        f, pq_centers=ivf.pq.centers, sqrt_n_blocks=ivf.pq.sqrt_n_blocks, **extra,
        active_centers=ivf.active_centers, center_size=ivf.pq_transformed_centers.size,
        center_codes=ivf.pq_transformed_centers.packed,
        list_sizes=np.array([ivf.pq_transformed_points[i].size for i in range(L)], np.int64),
        list_codes=np.concatenate([ivf.pq_transformed_points[i].packed for i in range(L)]),
        ids=np.concatenate([np.asarray(ivf.ids[i], np.int64) for i in range(L)])))
    return ivf, cent


def synth_rows_host(n, d, seed, cent, sigma, row0=0):
    """Rows of the seeded device generator (devbuild.hip), copied to the host."""
    from tinyknn_amd import _lib
    out = np.zeros((n, d), dtype=np.float32)
    c = None if cent is None else np.ascontiguousarray(cent, dtype=np.float32)
    _lib.check(_lib.lib().tk_synth_rows(_lib.ptr(out, _lib._f32p), row0, n, d, seed,
                                        None if c is None else c.ctypes.data, 0 if c is None else len(c),
                                        float(sigma)))
    return out


def build_index_c5(args, device):
    """BASELINE configs[4] on ONE GPU at full size: N = 100M x 128 float32 (51 GB), generated IN
    HBM by the seeded counter-based generator (3000 Gaussian clusters, sigma 0.7; SURVEY 8d C5:
    "per-GPU generation on device"), IVF n_clusters = 10 000 fitted on a 1M-row sample, PQ rotated
    128 -> 64 dims (M = 32, float64 table math), lists and codes built on the device
    (tk_index_build_dev).  The vectors never visit the host."""
    import torch
    from tinyknn_amd import IVF, FastPQ
    t0 = time.time()
    cent = np.random.RandomState(args.seed).randn(3000, args.d).astype(np.float32)
    sigma = 0.7
    micro = int(getattr(args, "c5_micro", 0) or 0)
    if micro > 0:
        # A second stand-in (round 6; the default stays what rounds 2-5 measured): isotropic noise of sigma 0.7 in all 128
        # dimensions around 3000 centres leaves a 128-bit PQ code nothing to tell a cluster's 33 000 rows apart by
        # (Recall10@10 0.195 at n_probes 10).  Here every cluster is `micro` micro-clusters (centre + 0.7 N(0, 1)) of
        # ~N / (3000 micro) rows each with a noise of --c5-sigma around them: local structure a code can resolve, as real
        # descriptor sets have — the same generator, kernels, sizes and list lengths, at a recall one would run at.
        off = np.random.RandomState(args.seed + 1).randn(3000 * micro, args.d).astype(np.float32)
        cent = (np.repeat(cent, micro, axis=0) + np.float32(0.7) * off).astype(np.float32)
        sigma = float(args.c5_sigma)
    args.c5_point_sigma = sigma
    ivf = IVF("euclidean", args.n_clusters, FastPQ(2))
    # The FIT (k-means on the GPU: atomics; ortho_group / k-means of FastPQ.fit: numpy's global RNG)
    # is not bit-reproducible from run to run, and a list-sharded index needs every rank to hold
    # the SAME centres and codebook: rank 0 fits first (main() orders the ranks) and leaves the
    # fitted parameters in the cache directory; the other ranks load them.  The BUILD from those
    # parameters is deterministic (seeded generator, stable sort) and runs on every rank.
    fit_cache = os.path.join(args.cache_dir, f"tinyknn_bench_c5fit_n{args.n}_d{args.d}_c{args.n_clusters}_s{args.seed}"
                             + (f"_m{micro}_g{sigma}" if micro > 0 else "") + ".npz")
    if os.path.exists(fit_cache):
        z = np.load(fit_cache)
        ivf.all_centers = z["all_centers"]
        ivf.pq.centers = z["pq_centers"]
        ivf.pq.sqrt_n_blocks = float(z["sqrt_n_blocks"])
        ivf.pq.R = z["R"] if "R" in z else None
        log(f"[bench] c5 fit loaded from {fit_cache}")
    else:
        ns = min(args.n, 1_000_000)
        sample = synth_rows_host(ns, args.d, args.seed, cent, sigma)   # rows 0..ns: a uniform sample of the clusters
        ivf.all_centers = quick_kmeans(sample, args.n_clusters, 6, args.seed, device).astype(np.float32)
        ivf.pq.fit(sample[:30000])
        del sample
        torch.cuda.empty_cache()
        try:
            extra = {} if ivf.pq.R is None else {"R": ivf.pq.R}
            np.savez(fit_cache, all_centers=ivf.all_centers, pq_centers=ivf.pq.centers,
                     sqrt_n_blocks=ivf.pq.sqrt_n_blocks, **extra)
        except OSError as e:
            log(f"[bench] could not cache the c5 fit: {e}")
        log(f"[bench] c5 fit done in {time.time() - t0:.1f}s")
    ivf.build_resident(args.n, args.d, args.seed, cent, sigma)
    sz = ivf.list_sizes
    log(f"[bench] c5 index built on the device in {time.time() - t0:.1f}s: {len(sz)} lists of "
        f"{sz.min()}..{sz.max()} rows, codes {int(((sz + 15) // 16).sum()) * 16 * (ivf.pq.centers.shape[1] // 4) / 1e6:.0f} MB, "
        f"vectors {args.n * args.d * 4 / 1e9:.1f} GB in HBM")
    return ivf, cent


def oracle_index_resident(ivf, cache_dir):
    """CPU oracle over what a device-resident index exports.  The 51 GB of vectors stay in HBM:
    the oracle's `data` is a sparse memory-mapped file into which only the rows a check needs
    (the heap candidates of the sampled queries) are copied (fill_rows)."""
    from oracle import oracle as O
    dev = ivf.device_index()
    sizes, codes, ids = dev.export_lists()
    chunks = (sizes + 15) // 16
    coff = np.concatenate([[0], np.cumsum(chunks)])
    ioff = np.concatenate([[0], np.cumsum(sizes)])
    L = len(sizes)
    path = os.path.join(cache_dir, f"tinyknn_c5_rows_{os.getpid()}.f32")
    data = np.memmap(path, dtype=np.float32, mode="w+", shape=(dev.N, dev.d))
    ox = O.OracleIndex(ivf.pq.centers, 2, ivf.pq.R, ivf.pq.sqrt_n_blocks, ivf.active_centers,
                       ivf.pq_transformed_centers.packed, [codes[coff[i]:coff[i + 1]] for i in range(L)],
                       list(sizes), [ids[ioff[i]:ioff[i + 1]] for i in range(L)], data)
    assert ox.data is data or np.may_share_memory(ox.data, data), "the oracle copied the sparse vector file"

    def fill_rows(rows):
        rows = np.unique(rows[rows >= 0])
        data[rows] = dev.read_rows(rows)

    def cleanup():
        try:
            os.unlink(path)
        except OSError:
            pass
    return ox, fill_rows, cleanup


def oracle_index(ivf):
    from oracle import oracle as O
    L = len(ivf.active_centers)
    return O.OracleIndex(ivf.pq.centers, 2, ivf.pq.R, ivf.pq.sqrt_n_blocks, ivf.active_centers,
                         ivf.pq_transformed_centers.packed,
                         [ivf.pq_transformed_points[i].packed for i in range(L)],
                         [ivf.pq_transformed_points[i].size for i in range(L)],
                         [ivf.ids[i] for i in range(L)], ivf.data)


def shard_inputs(args, ivf, cent, dev, device):
    """The shared batch of the sharded legs on the device + the unsharded index's rows for it."""
    import torch
    qs = synth_queries(cent, args.nq, args.seed + 100, kind=args.data)     # rank 0's batch
    qn, qp = ivf._prepare(qs.copy())
    qn_t = torch.from_numpy(qn).to(device)
    qp_t = torch.from_numpy(np.ascontiguousarray(qp)).to(device)
    want = torch.full((args.nq, args.k), -1, dtype=torch.int64, device=device)
    dev.query_batch_dev(qn_t.data_ptr(), qp_t.data_ptr(), qp.dtype != np.float32, args.nq, args.k,
                        args.n_probes, want.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    dev.join(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return qn_t, qp_t, want.cpu().numpy()


def list_sharded_leg(args, ivf, cent, dev, device, world, rank):
    """Strong-scaling leg: ONE batch of args.nq queries per step, shared by all ranks;
    lists sharded by cluster id.  Returns a dict (rank 0 reports it)."""
    import torch
    import torch.distributed as dist
    from tinyknn_amd.multi_gpu import ListShardedIndex
    qn_t, qp_t, want = shard_inputs(args, ivf, cent, dev, device)
    # auto: six steps per exchange (a rank's home share is then at least most of a batch: the latency-bound
    # stages — two heap replays, the collectives — cost the same for 10 000 as for 60 000 home queries;
    # same box, W = 1: 3 steps 18.5 M queries/s, 4: 19.4 M, 6: 21.1 M, profiles/r05/shard_w1_coalesce_depth.txt)
    co = args.shard_coalesce if args.shard_coalesce > 0 else max(6, world)
    co = max(1, min(co, 131072 // args.nq))
    idx = ListShardedIndex(ivf, depth=args.shard_depth, coarse=args.shard_coarse, coalesce=co,
                           force_collectives=args.force_collectives, counts=args.shard_counts,
                           plain={0: False, 1: True, 2: "two-phase", 3: "head"}[args.shard_plain])
    if args.shard_exchange == "auto":
        idx.exchange = "auto"          # ListShardedIndex._exchange_kind: filtered where lists are long
        kinds = [idx._exchange_kind(args.k, args.n_probes, None)]
    else:
        kinds = ["dense", "filtered"] if args.shard_exchange == "both" else [args.shard_exchange]
    res = None
    for kind in kinds:
        idx.exchange = kind        # same shard of the index, same buffers; only the exchange differs
        r = _list_sharded_run(args, idx, device, world, rank, qn_t, qp_t, want, co, kind)
        if res is None:
            res = r
        else:
            res["filtered_exchange"] = {k_: r[k_] for k_ in ("queries_per_s", "ms_per_step",
                                                             "identical_rows_vs_replica", "exchange")}
    # the same with ONE step per exchange (fixed Q = --nq queries per exchange: SURVEY 8e's definition
    # of strong scaling), beside the coalesced figure above
    if co > 1:
        idx.exchange = kinds[0]
        idx.coalesce = 1
        r1 = _list_sharded_run(args, idx, device, world, rank, qn_t, qp_t, want, 1, kinds[0])
        res["fixed_q_per_exchange"] = {k_: r1[k_] for k_ in ("queries_per_s", "ms_per_step",
                                                              "identical_rows_vs_replica", "steps_coalesced_per_exchange")}
        idx.coalesce = co
    res["streams"] = ("by role: front (coarse stages, probe all-gather) / scan (every scan, in order) / two replay "
                      "streams (exchange, replay, rescoring, id gather)" if idx._roles is not None and kinds[0] == "dense"
                      else "one stream per batch in flight")
    if world == 1 and args.rank_share > 1 and getattr(ivf, "pq_transformed_points", 0) is not None:
        # (a resident index was sharded IN PLACE by the leg above: main() runs its rank share first)
        del idx
        torch.cuda.empty_cache()
        try:
            res["rank_share_W%d" % args.rank_share] = rank_share_leg(args, ivf, device, qn_t, qp_t, want, args.rank_share)
        except Exception as e:      # noqa: BLE001 - an extra leg must not lose the line
            res["rank_share_W%d" % args.rank_share] = {"error": repr(e)}
    return res


def rank_share_leg(args, ivf, device, qn_t, qp_t, want, W=8, co=0):
    """What ONE GPU can prove about W ranks: the time of ONE rank's share of a W-rank list partition —
    its 1/W of the lists for all queries of the shared batches, the coarse stage, replay and rescoring
    of its home queries, every exchange buffer filled locally with what the other W - 1 ranks
    contribute (recorded from clone shards on this device: multi_gpu.SimulatedPeers; a device copy
    stands in for the bytes the links would land in this rank's HBM).  W consecutive steps are answered
    as one sharded batch (a rank's home share is then one batch of --nq queries), as the N > 1 leg does."""
    import torch
    from tinyknn_amd.multi_gpu import ListShardedIndex
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from simulated_peers import SimulatedPeers
    peers = SimulatedPeers(ivf, world=W, rank=0)
    try:
        co = max(1, min(co or max(W, args.shard_coalesce), 131072 // args.nq))
        idx = ListShardedIndex(ivf, simulate=peers, depth=args.shard_depth, coarse="home", coalesce=co,
                               counts="device", plain={0: False, 1: True, 2: "two-phase", 3: "head"}[args.shard_plain],
                               exchange="auto")
        kind = idx._exchange_kind(args.k, args.n_probes, None)
        if getattr(args, "rank_share_exchange", "auto") != "auto":
            kind = args.rank_share_exchange
        idx.exchange = kind
        r = _list_sharded_run(args, idx, device, 1, 0, qn_t, qp_t, want, co, kind, sim=peers)
        out = {k_: r[k_] for k_ in ("ms_per_step", "host_enqueue_ms_per_step", "identical_rows_vs_replica", "rows", "windows_ms",
                                    "window_drift_last_third_over_first_third", "windows_repeated_after_overflow",
                                    "timed_regions_repeated_after_a_peer_recording", "exchange", "scan",
                                    "code_chunks_per_rank", "batches_in_flight", "steps_coalesced_per_exchange")}
        if idx.stage_events:        # TINYKNN_SHARD_STAGE_EVENTS=1 with role streams: the last batches' stages, us
            torch.cuda.synchronize()
            evs = idx.stage_events[-10:]
            z = evs[0][0]
            out["stage_timeline_us"] = [[round(z.elapsed_time(e) * 1e3) for e in ev] for ev in evs]
            out["stage_timeline_is"] = "per batch: front start, front end, scan start, scan end, back end (us from the first)"
        # What the links would have to carry (xGMI: 7 links x ~153 GB/s per GPU, point to point): this rank's
        # INBOUND bytes per sharded batch — the other W - 1 ranks' regions of the equal-split all-to-all, the probe lists
        # and the ids — and the rate that needs at the period measured here WITHOUT links.
        ex = r["exchange"]
        per_step = (W - 1) / W * (ex["all_to_all_bytes_per_rank_per_step"] + ex["probe_all_gather_bytes_per_rank_per_step"]
                                  + ex["all_gather_bytes_per_rank_per_step"])
        out["links"] = {kind: {"inbound_MB_per_batch": per_step * co / 1e6,
                               "GBps_needed_at_the_measured_period": per_step / (r["ms_per_step"] * 1e-3) / 1e9,
                               "GBps_needed_at_0.7_efficiency": None}}
        if getattr(args, "rank_share_links", "dense") == "both" and kind == "dense" and args.workload != "c5":
            # ... and the filtered exchange beside it (records of the blocks below the bound after the first list,
            # fixed regions): fewer bytes, more kernels — decided by the bytes, not by the one-rank rate
            try:
                idx2 = ListShardedIndex(ivf, simulate=peers, depth=args.shard_depth, coarse="home", coalesce=co,
                                        counts="device", plain={0: False, 1: True, 2: "two-phase", 3: "head"}[args.shard_plain],
                                        exchange="filtered")
                peers.reset()
                r2 = _list_sharded_run(args, idx2, device, 1, 0, qn_t, qp_t, want, co, "filtered", sim=peers)
                e2 = r2["exchange"]
                # (regions travel whole: world x region records x 20 B per rank and exchange, + the counts)
                reg_bytes = W * e2.get("record_region", 0) * 20 / co
                ps2 = (W - 1) / W * (reg_bytes + e2["probe_all_gather_bytes_per_rank_per_step"] + e2["all_gather_bytes_per_rank_per_step"]
                                     + args.nq) + 0.0
                out["links"]["filtered"] = {"inbound_MB_per_batch": ps2 * co / 1e6, "ms_per_step": r2["ms_per_step"],
                                            "identical_rows_vs_replica": r2["identical_rows_vs_replica"],
                                            "GBps_needed_at_the_measured_period": ps2 / (r2["ms_per_step"] * 1e-3) / 1e9,
                                            "records_held_MB_per_batch": e2.get("records_held_bytes_per_rank_per_step", 0) * co / 1e6}
            except Exception as e:      # noqa: BLE001 - an extra figure must not lose the leg
                out["links"]["filtered"] = {"error": repr(e)}
        out.update(world=W, rank=0,
                   what="ONE rank's share of a W-rank partition on this GPU, the peers' contributions recorded and "
                        "copied in (no links): per step of --nq shared queries")
        return out
    finally:
        peers.close()


def _links_at_target(rs):
    """rank_share leg: the all-to-all rate each exchange would need on the links at 0.7 strong-scaling efficiency."""
    co = max(1, int(rs.get("steps_coalesced_per_exchange", 1)))
    for v in (rs.get("links") or {}).values():
        if isinstance(v, dict) and "inbound_MB_per_batch" in v:
            v["GBps_needed_at_0.7_efficiency"] = v["inbound_MB_per_batch"] * 1e6 / co / (rs["target_ms_per_step_at_0.7_efficiency"] * 1e-3) / 1e9


def _list_sharded_run(args, idx, device, world, rank, qn_t, qp_t, want, co, kind, sim=None):
    import torch
    import torch.distributed as dist
    lo_h, hi_h = (0, args.nq) if sim is None else sim.home_range(args.nq)
    # three looks per batch shape: the capacities and record regions are trimmed on the evidence of three
    # batches (multi_gpu.py) and are then final — nothing is re-sized inside the timed windows.
    # Rows checked against the unsharded answer: all of them (a simulated rank: its home rows).
    bad = 0
    for _ in range(3):
        got = idx.query_prepared(qn_t, qp_t, args.k, args.n_probes)
        bad = max(bad, int((got[lo_h:hi_h] != want[lo_h:hi_h]).any(axis=1).sum()))
    if co > 1:      # the coalesced batch: settles its capacity, and its rows must repeat `want`
        qn_c, qp_c = torch.cat([qn_t] * co), torch.cat([qp_t] * co)
        wc = np.concatenate([want] * co)
        lo_c, hi_c = (0, co * args.nq) if sim is None else sim.home_range(co * args.nq)
        for _ in range(3):
            gotc = idx.query_prepared(qn_c, qp_c, args.k, args.n_probes)
            bad = max(bad, int((gotc[lo_c:hi_c] != wc[lo_c:hi_c]).any(axis=1).sum()))
    rows_checked = (hi_h - lo_h) if co == 1 else (hi_c - lo_c)
    same = rows_checked - bad
    # every workspace slot must have seen a batch of the timed size before the clock starts: a slot
    # that grows inside the timed region pays hipMalloc/hipFree of gigabytes there (seen as a 4x
    # slower leg whenever the untimed calls above had left slot 0 at the single-batch size)
    host_s = [0.0, 0]

    def run(n):
        """n submits + join; join() raises if a batch in flight overflowed its exchange regions
        (it has grown them): the run is then repeated — never timed with invalid rows."""
        for attempt in range(4):
            try:
                th = time.perf_counter()
                for _ in range(n):
                    idx.submit(qn_t, qp_t, args.k, args.n_probes)
                host_s[0] += time.perf_counter() - th
                host_s[1] += n
                idx.join()
                torch.cuda.synchronize()
                return attempt
            except RuntimeError as e:
                if "submit the batches again" not in str(e):
                    raise
                log(f"[bench] rank {rank}: {e}")
        raise RuntimeError("list-sharded leg: batches had to be repeated four times in a row")

    run(-(-max(args.warmup, args.shard_depth * co) // co) * co)
    if world > 1:
        dist.barrier()
    # Timed region: ONE continuous run of `n_win` windows, as the replica region — a window is the smallest
    # multiple of `co` steps >= --steps (whole sharded batches), closed by an event on a side stream that
    # waits for every batch submitted so far (batches in flight finish out of order); nothing is
    # drained and no submission waits (the collectives keep the ranks in step); ms_per_step = the
    # median window behind the first (which pays the pipeline's fill).  A
    # region in which a batch had to be repeated (overflow, failed plain check: join() raises) is run again.
    steps_w = -(-max(args.steps, 6 * co if sim is not None else args.steps) // co) * co    # (a simulated rank of W: >= 6 batches per window)
    n_win = args.windows if args.windows > 0 else max(4, min(10, -(-300 // steps_w) + 1))
    idx.bytes_sent = idx.bytes_dense = 0
    repeats = 0
    timing_stream = torch.cuda.Stream()
    peer_records = 0
    for attempt in range(4):
        rec0 = sim.records if sim is not None else 0
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_win + 1)]
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        host_s[0], host_s[1] = 0.0, 0
        th = time.perf_counter()
        evs[0].record()
        for w in range(n_win):
            for _ in range(steps_w):
                idx.submit(qn_t, qp_t, args.k, args.n_probes)
            idx.flush_host_decisions()
            for bs in idx.batch_streams() or [torch.cuda.current_stream()]:
                timing_stream.wait_stream(bs)
            evs[w + 1].record(timing_stream)
        host_s[0], host_s[1] = time.perf_counter() - th, n_win * steps_w
        try:
            idx.join()
            torch.cuda.synchronize()
            if sim is not None and sim.records != rec0 and attempt < 3:
                # the simulated peers recorded their contributions again INSIDE the timed region (the batch changed
                # its form: a recording pass runs all W ranks once, a second or more) — not the live rank's time
                peer_records += sim.records - rec0
                log(f"[bench] simulated peers recorded {sim.records - rec0} time(s) inside the timed region: run again")
                continue
            break
        except RuntimeError as e:
            if "submit the batches again" not in str(e) or attempt == 3:
                raise
            log(f"[bench] rank {rank}: {e}")
            repeats += 1
            torch.cuda.synchronize()
    wins_all = [evs[w].elapsed_time(evs[w + 1]) * 1e-3 for w in range(n_win)]
    t = torch.tensor(wins_all, dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wins_all = t.tolist()           # in time order; the first one pays the fill
    wins = wins_all[1:]
    # the MEAN window behind the first: with several batches in flight on their own streams the device
    # does not finish batches in submission order, so single windows swing (a median would pick a
    # lucky or an unlucky one); their sum is simply the time the steps behind the fill took
    el = sum(wins) / len(wins)
    third = max(1, len(wins) // 3)
    drift = (sum(wins[-third:]) / third) / (sum(wins[:third]) / third)
    cap = idx.capacity[(args.nq * co, args.n_probes)] if (args.nq * co, args.n_probes) in idx.capacity \
        else idx.capacity[(args.nq, args.n_probes)]
    W_ix = idx.world            # (a simulated rank: the partition's world, not the process group's)
    load = np.bincount(idx.owner, weights=(idx.list_sizes + 15) // 16, minlength=W_ix)
    filt = {"kind": "dense (whole segments at fixed positions, no host synchronisation)"}
    if kind == "filtered":
        # measured in the timed steps, this rank: records (20 B per block that travels) + bounds +
        # counts, against the blocks of whole segments (16 B each, what the dense form needs at least)
        dense_b = idx.bytes_dense
        filt = {"kind": "filtered (bound after the first probed list; blocks below it as 20-byte records)",
                "record_bytes_per_rank_per_step": int(idx.bytes_sent // (steps_w * n_win)),
                "whole_segment_bytes_per_rank_per_step": int(dense_b // (steps_w * n_win)),
                "bytes_ratio": round(idx.bytes_sent / max(1, dense_b), 4),
                "host_syncs_per_exchange": 0 if idx.counts == "device" else 1,
                "counts": idx.counts + (" (fixed record regions, equal-split all-to-all, counts read on the device)"
                                        if idx.counts == "device" else " (variable splits read on the host)")}
        if idx.counts == "device":
            key = (args.nq * co, args.n_probes)
            filt["record_region"] = int(idx._region(key[0], key[1], idx._capacity(*key)))
            filt["records_held_bytes_per_rank_per_step"] = int(20 * idx.records_sent // (steps_w * n_win))
    return {"queries_per_s": args.nq * steps_w / el, "ms_per_step": el / steps_w * 1e3,
            "steps_per_window": steps_w,
            "timing": "one continuous run; a window = %d steps (whole sharded batches), closed by an event behind every "
                      "batch submitted so far; MEAN of the windows behind the first (= elapsed / steps behind the fill)" % steps_w,
            "scaling": "strong (one shared batch of %d queries per step)" % args.nq,
            "identical_rows_vs_replica": same, "rows": rows_checked,
            "host_enqueue_ms_per_step": host_s[0] / max(host_s[1], 1) * 1e3,
            "windows_ms": [w_ * 1e3 for w_ in wins_all], "windows_in_time_order": True,
            "window_drift_last_third_over_first_third": drift, "windows_repeated_after_overflow": repeats,
            "timed_regions_repeated_after_a_peer_recording": peer_records,
            "exchange": {**filt, "all_to_all_bytes_per_rank_per_step": int(W_ix * cap * 16 // co),
                         "region_capacity_uint4": int(cap),
                         "probe_all_gather_bytes_per_rank_per_step":
                             int(-(-args.nq // W_ix) * min(args.n_probes, len(idx.list_sizes)) * 8)
                             if args.shard_coarse == "home" else 0,
                         "all_gather_bytes_per_rank_per_step": int((-(-args.nq // W_ix) * args.k + 1) * 8)},
            "coarse_stage": args.shard_coarse + (" (tables for all queries, coarse scan/replay/rescoring of the "
                                                 "rank's nq/W home queries, probe lists all-gathered)"
                                                 if args.shard_coarse == "home" else " on every rank"),
            "scan": ({"form": "one phase, as the unsharded pipeline: heads of the first lists exactly, everything else on "
                              "the int8 matrix cores, the home replay checks the lemma per query and flags the batch "
                              "(tk_index_shard_scan_plain_dev); no bound exchange",
                      "arguments_switched_to_two_phase": [list(map(str, a)) for a in sorted(idx._plain_failed, key=str)]}
                     if idx._one_phase_now(args.k, args.n_probes, None) else
                     {"form": "one phase behind ONE byte per query: heads of the first lists exactly + the bound after them "
                              "(its owner replays, MIN all-reduce), then heads exactly / everything else on the int8 matrix "
                              "cores for the queries whose bound is at most their table's limit, exactly for the others "
                              "(tk_index_shard_scan_head_dev / _scan_plain_dev(bound)): the check at home cannot fail"}
                     if idx._head_phase_now(args.k, args.n_probes, None) else
                     {"form": "two-phase: first probed lists exactly, bound min-reduced over the ranks, the lists behind "
                              "them on the int8 matrix cores where the bound allows (tk_index_shard_scan_first_dev / _rest_dev)",
                      "last_batch_this_rank": idx.engine.dev.shard_plain_stats((idx._calls - 1) % idx.depth)}
                     if idx._use_plain(args.k, args.n_probes, None) else
                     {"form": "one phase, exact kernel (tk_index_shard_scan_dev)"}),
            "code_chunks_per_rank": [int(x) for x in load], "batches_in_flight": args.shard_depth,
            "steps_coalesced_per_exchange": co}


def raw_stream_leg(args, dev, qs, out_dev, device, world):
    """RAW queries on the host in, ids on the host out: what a caller of the drop-in API gets.
    Every step = the exact host preparation of ivf.py:125-128 (numpy's own BLAS calls on a
    thread pool, tinyknn_amd/_front.py), pinned H2D, the device pipeline, D2H of the ids —
    all inside the timed region, batches overlapped by a streaming session (tk_stream_*)."""
    import torch
    import torch.distributed as dist
    from tinyknn_amd import _front
    if not _front.bind():
        return {"error": "numpy's BLAS could not be bound: " + str(_front.info()["why"])}
    # a step = one batch of args.nq queries, submitted in pieces of at most one sub-batch of the
    # index (distance rows of a sub-batch stay under ~4 GiB: long lists mean fewer queries)
    piece = min(args.nq, dev.max_sub_batch(args.k, args.n_probes))
    cuts = list(range(0, args.nq, piece))
    # (pairs of submits run as one batch when the index coalesces: twice the tickets keep as many batches in flight)
    slots = 8 * len(cuts) * launch_batches(args)
    st = dev.stream(piece, args.k, args.n_probes, slots=min(slots, 64))
    nbuf = max(2, min(slots, 64) // len(cuts))
    outs = [np.full((args.nq, args.k), -1, dtype=np.int64) for _ in range(nbuf)]
    qparts = [np.ascontiguousarray(qs[c:c + piece]) for c in cuts]

    def submit(i):
        o = outs[i % nbuf]
        t = None
        for c, qp_ in zip(cuts, qparts):
            t = st.submit(qp_, o[c:c + piece])
        return t

    for i in range(max(args.warmup, nbuf)):
        submit(i)
    st.drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    # a short run first (--steps steps between two drains: what round 5 reported — the pipeline's fill and drain, ~1.5 ms,
    # are a fifth of it), then the leg's figure: ONE continuous run of 15 x --steps steps, as the headline's timed region
    # is, fill and drain INSIDE the time (1-2 % of it)
    t0 = time.perf_counter()
    for i in range(args.steps):
        submit(i)
    st.drain()
    torch.cuda.synchronize()
    el_short = time.perf_counter() - t0
    n_long = 15 * args.steps
    if world > 1:
        dist.barrier()
    p0 = st.prepare_seconds()
    t0 = time.perf_counter()
    for i in range(n_long):
        submit(i)
    st.drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    prep = (st.prepare_seconds() - p0) / n_long
    # one batch at a time, for the latency of a single call
    t1 = time.perf_counter()
    for i in range(5):
        submit(0)
        st.drain()
    lat = (time.perf_counter() - t1) / 5
    st.close()
    t = torch.tensor([el], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    ref = out_dev.cpu().numpy()
    fi = _front.info()
    return {"queries_per_s": args.nq * world * n_long / el, "ms_per_step": el / n_long * 1e3,
            "steps": n_long, "timing": "one continuous run between two drains, fill and drain of the pipeline included",
            "short_run": {"steps": args.steps, "queries_per_s": args.nq * args.steps / el_short,
                          "note": "--steps steps between two drains (rank 0): the fill and drain of the pipeline are a "
                                  "fifth of such a run; this was the leg's figure until round 5"},
            "rows_identical_to_device_resident_path": int(min((o == ref).all(axis=1).sum() for o in outs)),
            "rows": args.nq,
            "host_prepare_ms_per_batch": prep * 1e3, "host_threads": fi["threads"],
            "host_blas": os.path.basename(fi["path"]),
            "single_batch_latency_ms": lat * 1e3,
            "note": "raw float32 queries (host) -> ids (host), EXACT: normalisation/rotation by numpy's own "
                    "cblas_sdot/cblas_dgemv on a thread pool, pinned async H2D/D2H on a copy stream, "
                    "8 batches outstanding; preparation + copies + kernels all inside the timed region",
            "queries_per_submit": piece}


def hbm_scale_leg(device):
    """The code scan where it IS HBM-bound: 1 GiB of random packed codes (M = 32, 2^26 codes —
    four times the 256 MiB Infinity Cache), every byte streamed once per pass.  nq = 1: the
    query-major kernel, algorithmic bytes == fetched bytes; nq = 4 / 16: the list-major kernel
    (each code byte fetched once per 4 queries)."""
    import torch
    from tinyknn_amd import _lib
    L = _lib.lib()
    stream = torch.cuda.current_stream().cuda_stream
    M, n = 32, 1 << 26
    chunks = n // 16
    rng = np.random.default_rng(0)
    packed = rng.integers(0, 2**63, size=(chunks, M), dtype=np.int64).view(np.uint64)
    h = L.tk_codes_upload(_lib.ptr(packed, _lib._u64p), chunks, M)
    del packed
    if not h:
        return {"error": L.tk_last_error().decode()}
    res = {"code_bytes": n * M // 2, "M": M, "codes": n, "cases": []}
    try:
        for nq in (1, 4, 16):
            tables = rng.integers(-4, 24, size=(nq, M, 16)).astype(np.int8).view(np.uint8)
            t_dev = torch.from_numpy(tables).to(device)
            out = torch.empty((nq, chunks * 16), dtype=torch.uint8, device=device)
            run = lambda: _lib.check(L.tk_codes_estimate_dev(h, t_dev.data_ptr(), nq, out.data_ptr(), 1, 1, stream))
            run()
            torch.cuda.synchronize()
            reps = 20 if nq == 1 else 8
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            alg = nq * n * (M // 2) + nq * n          # code bytes per (query, code) + int8 out
            hbm = -(-nq // 4) * n * (M // 2) + nq * n if nq >= 4 else alg     # bytes that must cross HBM
            res["cases"].append({"kernel": "scan_units_kernel (list-major)" if nq >= 4 else "scan_flat_kernel (query-major)",
                                 "nq": nq, "ms": ms, "algorithmic_GBps": alg / ms / 1e6,
                                 "min_hbm_GBps": hbm / ms / 1e6,
                                 # a fraction of the HBM peak only from bytes that must cross HBM
                                 "frac": hbm / ms / 1e6 / HBM_PEAK_GBPS})
            del out
    finally:
        L.tk_codes_free(h)
    c1 = res["cases"][0]
    res["GBps"], res["frac"] = c1["algorithmic_GBps"], c1["frac"]
    res["note"] = ("GBps/frac: nq = 1, where every code byte is fetched from HBM exactly once per launch "
                   "(algorithmic = real traffic); HIP events on the launch stream")
    return res


class ctypes_double:
    """a C double and its byref (tk_measure_* out-parameters)"""

    def __init__(self):
        import ctypes
        self._v = ctypes.c_double(0.0)
        self.ref = ctypes.byref(self._v)

    @property
    def value(self):
        return self._v.value


def launch_batches(args):
    """Steps whose queries ONE scan launch of the timed region covers (tk_index_set_coalesce): two,
    where a pair's distance rows fit one workspace (main() sets args.pairs_fit)."""
    return 2 if (args.coalesce == 2 and args.pipeline > 1 and getattr(args, "pairs_fit", True)) else 1


def kernel_stats_child(args, plain_on=True):
    """Per-kernel durations of the TIMED configuration: a child run of this script's timed region under
    `rocprofv3 --kernel-trace --stats` (same workload, pipeline depth and coalescing; 50 steps per
    window).  Returns {kernel name: {"calls", "avg_us"}} and the path of the kept CSV."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not found"
    me = os.path.abspath(__file__)
    tmp = tempfile.mkdtemp(prefix="tk_kt_")
    cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", tmp, "--", sys.executable, me,
           "--steps", "50", "--warmup", "5", "--profile-only", "--shard", "none", "--traffic", "none", "--no-hbm-leg",
           "--no-cpu", "--cache-dir", args.cache_dir, "--pipeline", str(args.pipeline), "--coalesce", str(args.coalesce),
           "--workload", args.workload, "--n", str(args.n), "--d", str(args.d),
           "--n-clusters", str(args.n_clusters), "--nq", str(args.nq), "--k", str(args.k),
           "--n-probes", str(args.n_probes), "--metric", args.metric, "--data", args.data,
           "--build-probes", str(args.build_probes), "--seed", str(args.seed)]
    try:
        cp = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE,
                            stderr=subprocess.DEVNULL, timeout=900, check=True, text=True)
        fs = glob.glob(os.path.join(tmp, "**", "*kernel_stats.csv"), recursive=True)
        if not fs:
            return None, "no kernel_stats.csv in the child run"
        out = {}
        try:        # the child's own line: its step time and HIP-event kernel time UNDER the profiler
            cj = json.loads([l for l in cp.stdout.splitlines() if l.startswith("{")][-1])
            kernel_stats_child.own = {"ms_per_step": cj.get("ms_per_step"), "plain_kernel_ms_hip_events": cj.get("plain_kernel_ms")}
        except Exception:      # noqa: BLE001
            kernel_stats_child.own = None
        for r in csv.DictReader(open(fs[0])):
            out[r["Name"]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                              "pct": float(r["Percentage"])}
        keep = os.path.join(ROOT, "gpurun_out", "bench")
        os.makedirs(keep, exist_ok=True)
        kept = os.path.join(keep, "rocprofv3_kernel_stats_timed_region.csv")
        shutil.copy(fs[0], kept)
        return out, os.path.relpath(kept, ROOT)
    except Exception as e:     # noqa: BLE001 - profiling is an extra
        return None, f"kernel-trace child run failed: {e!r}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pick_kernel(stats, sub):
    """(name, entry) of the busiest kernel whose name contains `sub`."""
    if not stats:
        return None, None
    hits = [(n, e) for n, e in stats.items() if sub in n]
    if not hits:
        return None, None
    return max(hits, key=lambda t: t[1]["calls"] * t[1]["avg_us"])


def measure_traffic(args, plain_on=True):
    """HBM bytes per scan launch from PMC counters, measured NOW: two child runs of this script
    under rocprofv3 (--pmc FETCH_SIZE, then --pmc WRITE_SIZE SQ_INSTS_VALU: the two sizes do not fit one pass), one
    batch in flight so that the list scan and the coarse scan are separate launches.  gfx950
    correction of MI355X_MICROARCH.md (HBM): FETCH_SIZE reports half the bytes of wide coalesced
    reads -> x2; WRITE_SIZE exact; both in KiB."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not found"
    me = os.path.abspath(__file__)
    mode = {"TINYKNN_PLAIN_SCAN": "2"} if plain_on else {"TINYKNN_PLAIN_SCAN": "0"}

    def one_pass(counters):
        """One child run under --pmc `counters`: {counter: {scan launch: mean value per launch}}."""
        tmp = tempfile.mkdtemp(prefix="tk_pmc_")
        cmd = ["rocprofv3", "--pmc", *counters, "--output-format", "csv", "-d", tmp, "--", sys.executable, me,
               "--steps", "3", "--warmup", "1", "--warmup-seconds", "0", "--windows", "1", "--pipeline", "1",
               "--profile-only", "--shard", "none", "--cache-dir", args.cache_dir,
               "--workload", args.workload, "--n", str(args.n), "--d", str(args.d),
               "--n-clusters", str(args.n_clusters), "--nq", str(args.nq * launch_batches(args)), "--k", str(args.k),
               "--n-probes", str(args.n_probes), "--metric", args.metric, "--data", args.data,
               "--build-probes", str(args.build_probes), "--seed", str(args.seed)]
        try:
            # the child answers 4 batches: too few for the plain path to prove itself (probe, wait, on).
            # It runs in the state the parent's timed region settled in — always plain, or never.
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", **mode), stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=600, check=True)
            acc = {c: {} for c in counters}
            for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    name = r["Kernel_Name"]
                    # the scan launches of one batch in flight: the plain kernel (matrix cores), the
                    # exact list-major launch (heads + whole lists), the coarse scan
                    key = ("plain" if "scan_plain" in name else
                           "units2" if "scan_units2_kernel" in name else
                           "units" if "scan_units_kernel" in name else None)
                    if r["Counter_Name"] in acc and key:
                        acc[r["Counter_Name"]].setdefault(
                            key + ("_coarse" if key == "units" and
                                   name.split("(")[0].rstrip(">").rstrip().endswith("true") else ""),
                            []).append(float(r["Counter_Value"]))
            return {c: {k_: sum(v) / len(v) for k_, v in a.items()} for c, a in acc.items()}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)

    # FETCH_SIZE takes 3 of the L2's 4 counters and WRITE_SIZE 2: two passes.  SQ_INSTS_VALU (an extra: the
    # traffic figure stands without it) is a counter of another block and rides with WRITE_SIZE.
    tot = {"SQ_INSTS_VALU": None}
    measure_traffic.parts = {}
    for counters in (("FETCH_SIZE",), ("WRITE_SIZE", "SQ_INSTS_VALU")):
        try:
            got = one_pass(counters)
            if not got[counters[0]] and len(counters) > 1:
                got = one_pass(counters[:1])
        except Exception as e:     # noqa: BLE001 - profiling is an extra
            if len(counters) == 1:
                return None, f"{counters[0]} pass failed: {e!r}"
            try:
                got = one_pass(counters[:1])
            except Exception as e2:     # noqa: BLE001
                return None, f"{counters[0]} pass failed: {e2!r}"
        if not got[counters[0]]:
            return None, f"no scan kernel rows in the {counters[0]} pass"
        for c, parts in got.items():
            if parts:
                tot[c] = sum(parts.values())     # every scan launch of a batch (KiB for the two sizes)
                measure_traffic.parts[c] = parts
    measure_traffic.valu_insts = tot["SQ_INSTS_VALU"]      # wave-instructions, list + coarse scan
    return (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024, \
        "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this script in this run (pipeline 1: plain " \
        "kernel + exact list-major launch + coarse scan summed; FETCH x2 per the gfx950 correction)"


VALU_RATE_PER_SIMD = 0.52e9      # wave-instructions/s/SIMD of the scan's VOP3/VOP3P mix, measured:
                                 # profiles/r02_valu_issue_rate_microbench.txt (4.0-4.6 cycles each)


I8_MFMA_PEAK_TOPS = 5000.0      # dense int8 MFMA: 2x the bf16 rate per clock (MI355X_MICROARCH.md, Matrix cores)


def plain_roofline(st, M, nq):
    """The plain-sum scan (plain_scan.hip): its matrix-core work per batch.  A unit = 32 pairs x a
    list's chunk pairs; per chunk pair M/2 v_mfma_i32_32x32x32_i8 = M/2 x 32768 multiply-adds
    (one-hot(code) x table: 15 of 16 products are by zero — this is matrix-core OCCUPANCY, the
    useful work is the 26 B per (query, code) of `achieved`)."""
    if not st or not st["plain_units"]:
        return None
    mfma = st["plain_unit_chunk_pairs"] * (M // 2)
    floor = mfma * 32 / (1024 * 2.4e9) * 1e3
    return dict(st, mfma_instructions_per_batch=mfma, int8_ops_per_batch=mfma * 2 * 32768,
                mfma_floor_ms=floor,
                peak_unit="TOP/s int8 dense",
                peak=I8_MFMA_PEAK_TOPS, tile_fill=st["plain_pairs"] / (32.0 * st["plain_units"]),
                flagged_fraction=st["flagged_queries"] / float(nq),
                note="mfma_floor_ms: the batch's MFMA instructions at 32 cycles each on 1024 SIMDs at 2.4 GHz; "
                     "tile_fill: pairs per 32-pair tile; flagged: queries whose bound at the first plain block "
                     "was above the table's limit (scanned again exactly, replayed again)")


def timed_rate(dev, batches, qp_is_f64, nq, k, n_probes, stream, pipeline, coalesce, steps=40, windows=4):
    """queries/s of the pipelined mode for one (index, n_probes): one continuous run of `windows` x
    `steps` steps, every window closed by the completion event of its last batch; the median window
    behind the first (the protocol of the headline figure, shorter)."""
    import torch
    dev.set_pipeline(pipeline)
    dev.set_coalesce(coalesce if pipeline > 1 else 1)
    dev.reserve(nq * (2 if coalesce == 2 and pipeline > 1 else 1), k, n_probes)
    n = [0]

    def step(ev=None):
        b = batches[n[0] % len(batches)]
        n[0] += 1
        dev.query_batch_dev(b["q_dev"].data_ptr(), b["qp_dev"].data_ptr(), qp_is_f64, nq, k, n_probes,
                            b["out"].data_ptr(), stream=stream, done_event=ev)

    for _ in range(48):      # (clocks and every workspace of the pipeline settle)
        step()
    dev.join(stream)
    torch.cuda.synchronize()
    n[0] = 0
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(windows + 1)]
    for e in evs:
        e.record()
    torch.cuda.synchronize()
    evs[0].record()
    for w in range(windows):
        for i in range(steps):
            step(evs[w + 1].cuda_event if i == steps - 1 else None)
    dev.join(stream)
    torch.cuda.synchronize()
    ms = sorted(evs[w].elapsed_time(evs[w + 1]) for w in range(1, windows))
    per_step = ms[len(ms) // 2] / steps
    return {"queries_per_s": nq / (per_step * 1e-3), "ms_per_step": per_step, "steps_per_window": steps,
            "windows": windows}


def sweep_leg(args, ivf, dev, batches, qp_is_f64, stream, device, cent):
    """BASELINE configs[3] and the reference's own sweep (examples/bench.py:108-139) in front of the
    driver: n_probes 1 / 5 / 20 / 50 on the headline index, then the reference's default build
    (IVF.build(n_probes=2), ivf.py:53) and the SIFT-shaped stand-in of configs[2] at n_probes 10 — each
    with its rate (timed_rate) and >= 500 result rows compared with the CPU oracle."""
    import torch
    rows = 500
    out = {"protocol": "pipelined mode, %d steps x 4 windows per point, median window behind the first; parity: the "
                       "first %d rows of batch 0 against oracle/tinyknn_oracle.c" % (40, rows), "points": []}

    oxs = {}

    def point(label, ivf_, dev_, batches_, f64, n_probes, extra=None):
        e = {"config": label, "n_probes": n_probes}
        try:
            e.update(timed_rate(dev_, batches_, f64, args.nq, args.k, n_probes, stream, args.pipeline, args.coalesce))
            torch.cuda.synchronize()
            got = batches_[0]["out"].cpu().numpy()[:rows]
            ox = oxs.get(id(ivf_)) or oxs.setdefault(id(ivf_), oracle_index(ivf_))
            tc = time.perf_counter()
            want = ox.query_batch(batches_[0]["qn"][:rows], args.k, n_probes)
            e["cpu_oracle_queries_per_s"] = rows / (time.perf_counter() - tc)
            e["parity_vs_oracle"] = {"queries_checked": rows, "identical_rows": int((want == got).all(axis=1).sum())}
            if extra:
                e.update(extra(dev_))
        except Exception as ex:     # noqa: BLE001 - a sweep point must not lose the line
            e["error"] = repr(ex)
        out["points"].append(e)

    for np_ in (1, 5, 20, 50):
        point("headline index", ivf, dev, batches, qp_is_f64, np_)
    dev.set_pipeline(1)
    lap("side roofs; sweep: n_probes 1 / 5 / 20 / 50 on the headline index")

    def other_index(label, **over):
        a2 = argparse.Namespace(**dict(vars(args), **over))
        t0 = time.perf_counter()
        ivf2, cent2 = build_index(a2, device)
        dev2 = ivf2.device_index()
        bs = []
        for b in range(2):
            qs_b = synth_queries(cent2, a2.nq, a2.seed + 100 + 1000 * b, kind=a2.data)
            qn_b, qp_b = ivf2._prepare(qs_b.copy())
            bs.append(dict(qn=qn_b, q_dev=torch.from_numpy(qn_b).to(device),
                           qp_dev=torch.from_numpy(np.ascontiguousarray(qp_b)).to(device),
                           out=torch.full((a2.nq, a2.k), -1, dtype=torch.int64, device=device)))
        f64 = qp_b.dtype != np.float32
        point(label + " (fit + build %.0f s)" % (time.perf_counter() - t0), ivf2, dev2, bs, f64, 10,
              extra=lambda d_: {"plain_scan_state": (d_.plain_stats() or {}).get("state")})
        dev2.set_pipeline(1)
        dev2.close()
        oxs.clear()
        lap("sweep: " + label[:40])

    try:
        other_index("IVF.build(n_probes=2): the reference's default build (ivf.py:53), every point in two lists",
                    build_probes=2)
        other_index("sift-clustered euclidean 1M x 128 (BASELINE configs[2] stand-in: SIFT's value range, 300 clusters), "
                    "n_clusters 1000, PQ rotated to 64 dims (M = 32, float64 tables)",
                    data="sift-clustered", metric="euclidean", d=128, n=1000000, n_clusters=1000)
    except Exception as ex:     # noqa: BLE001
        out["error"] = repr(ex)
    return out


def roofline_entry(args, M, pst, plain_ms, stages, iso_stages, scan_bytes, n_prof, traffic, traffic_src,
                   kstats, kstats_src, copy_gbps, read_gbps):
    """The roofline of the dominant kernel of the timed region, with the resource that binds it.

    Plain path on (the default batch): `scan_plain_wave_kernel` — one-hot(code) x table on
    v_mfma_i32_32x32x32_i8.  Its bound is matrix-core issue: `achieved` = int8 operations per launch /
    the kernel's duration, `peak` = the dense int8 MFMA peak (2 x bf16, MI355X_MICROARCH.md), `frac` =
    their ratio = (MFMA instructions x 32 cycles / (1024 SIMDs x 2.4 GHz)) / duration.  The duration is
    measured twice and both are printed: HIP events around the kernel on the stream it is launched on, in
    the timed region (`kernel_ms`, what `achieved` uses), and the rocprofv3 --kernel-trace --stats
    average of a child run of the same command (`kernel_ms_rocprofv3`; CSV kept under gpurun_out/bench/,
    committed as profiles/r04/).  HBM is the bound by contract only: the contractual
    algorithmic rate stays as `algorithmic_GBps`, and what the fabric really carried is
    `hbm_frac_measured` = PMC traffic / duration / 8 TB/s.  Plain path off: the exact launch against its
    VALU-issue floor (profiles/r02_scan_forms.md)."""
    lb = launch_batches(args)
    scan_ms = stages["scan"]
    ex_name, ex = pick_kernel(kstats, "scan_units2_kernel") if args.pipeline > 1 else pick_kernel(kstats, "scan_units_kernel")
    valu_parts = (getattr(measure_traffic, "parts", {}) or {}).get("SQ_INSTS_VALU") or {}
    exact_valu = sum(v for k_, v in valu_parts.items() if k_.startswith("units"))
    exact = None
    if exact_valu:
        floor_ms = exact_valu / (1024 * VALU_RATE_PER_SIMD) * 1e3
        exact = {"kernel": ex_name, "bound": "valu-issue", "wave_instructions_per_launch": exact_valu,
                 "issue_rate_per_simd": VALU_RATE_PER_SIMD, "floor_ms": floor_ms,
                 "kernel_ms_rocprofv3": None if not ex else ex["avg_us"] / 1e3,
                 "frac": None if not ex else floor_ms / (ex["avg_us"] / 1e3),
                 "note": "SQ_INSTS_VALU (PMC child run, one batch in flight, %d queries) of the exact launches — heads of "
                         "the first lists + whole lists of the queries that stay exact + the coarse scan — at the issue "
                         "rate measured for this instruction mix (profiles/r02_valu_issue_rate_microbench.txt)"
                         % (args.nq * lb)}
    if exact and exact["frac"] and exact["frac"] > 1.0:
        # the PMC child run and the kernel-trace child run did not launch the same thing (a batch cut into sub-batches:
        # the trace's average is over launches of different sizes) — no fraction is formed from the two
        exact["frac_not_formed"] = ("instructions of one launch of the PMC run against the average duration of the "
                                    "trace run's launches, which differ in size: %.2f" % exact["frac"])
        exact["frac"] = None
    common = {"traffic_source": traffic_src, "device_copy_GBps_measured": copy_gbps,
              "device_read_GBps_measured": read_gbps, "launches_timed": n_prof,
              "launch_covers": "%d step(s) = %d queries" % (lb, args.nq * lb),
              "algorithmic_bytes_per_launch": scan_bytes,
              "scan_stage_ms": scan_ms,
              "algorithmic_GBps": scan_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else None,
              "algorithmic_note": "SURVEY 8d's contractual figure: one code byte per (query, stored code) pair (+ table + "
                                  "heap bytes) of the batch over its scan stage (plain kernel + exact launch, HIP "
                                  "events on the scan stream).  It exceeds what HBM carries by the reuse factor (32 "
                                  "queries share each fetched tile; codes sit in L2 / Infinity Cache), so NO fraction "
                                  "of the HBM peak is formed from it",
              "exact_launch": exact,
              "kernel_stats_rocprofv3": None if not kstats else
              {n[:96]: e for n, e in sorted(kstats.items(), key=lambda t: -t[1]["calls"] * t[1]["avg_us"])[:12]},
              "kernel_stats_csv": kstats_src}
    if pst and pst.get("plain_units") and plain_ms > 0:
        pr = plain_roofline(pst, M, args.nq * lb)
        name, e = pick_kernel(kstats, "scan_plain")
        ops = pr["int8_ops_per_batch"]
        achieved = ops / (plain_ms * 1e-3) / 1e12
        ptraffic = None
        parts = getattr(measure_traffic, "parts", {}) or {}
        if traffic is not None and "plain" in (parts.get("FETCH_SIZE") or {}):
            ptraffic = (2 * parts["FETCH_SIZE"]["plain"] + parts["WRITE_SIZE"].get("plain", 0.0)) * 1024
        return dict(common, bound="mfma",
                    bound_measured="matrix-core issue of the plain-sum kernel (one MFMA per block pair and 32 x 32 tile; "
                                   "beside it the CU's texture path — table-row loads and output stores — is 67-77 % busy: "
                                   "profiles/r04/pmc_plain_kernel_micro.txt); the exact launch: VALU issue (`exact_launch`); "
                                   "hbm by contract only (`algorithmic_GBps`, `hbm_frac_measured`)",
                    kernel=name or "scan_plain_wave_kernel",
                    achieved=(achieved if not e else ops / (e["avg_us"] * 1e-6) / 1e12), peak=I8_MFMA_PEAK_TOPS,
                    unit="TOP/s",
                    frac=(achieved if not e else ops / (e["avg_us"] * 1e-6) / 1e12) / I8_MFMA_PEAK_TOPS,
                    frac_is="int8 operations of one launch / the kernel's average duration in the rocprofv3 "
                            "--kernel-trace --stats child run of this command (`kernel_ms_rocprofv3`, the kept CSV) / "
                            "peak = mfma_floor_ms / kernel_ms_rocprofv3.  `frac_hip_events_timed_region` is the same "
                            "ratio with the HIP-event duration of the un-profiled timed region (`kernel_ms`): there the "
                            "host enqueues faster, more kernels run beside the scan and each is longer; `child_run` "
                            "gives the profiled run's own HIP-event time — inside one run events and rocprofv3 agree",
                    kernel_ms=plain_ms, kernel_ms_source="HIP events in front of and behind the kernel on the scan stream, "
                                                         "timed region, every 4th batch (tk_index_last_profile)",
                    frac_hip_events_timed_region=achieved / I8_MFMA_PEAK_TOPS,
                    kernel_ms_rocprofv3=None if not e else e["avg_us"] / 1e3,
                    child_run=getattr(kernel_stats_child, "own", None),
                    mfma_floor_ms=pr["mfma_floor_ms"],
                    mfma_floor_note="MFMA instructions of the launch x 32 cycles / (1024 SIMDs x 2.4 GHz): the spec clock; "
                                    "inside this kernel the chip holds 1.8-2.1 GHz (s_memtime / s_memrealtime, "
                                    "profiles/r04/plain_kernel_clock.txt), at which the floor is 15-25 % higher",
                    traffic=ptraffic,
                    traffic_note="HBM bytes of ONE plain-kernel launch (PMC child run, FETCH x2 + WRITE, one batch of "
                                 "launch_covers queries in flight)",
                    hbm_frac_measured=None if ptraffic is None else ptraffic / (plain_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    all_scan_launches_traffic=traffic,
                    plain_scan=pr)
    # plain path off: the exact launch is the dominant scan kernel
    if exact and exact["frac"]:
        return dict(common, bound="valu", bound_measured="VALU issue of the exact scan's v_perm / packed-add mix "
                                                         "(profiles/r02_scan_forms.md); hbm by contract only",
                    kernel=ex_name, achieved=exact_valu / (exact["kernel_ms_rocprofv3"] * 1e-3) / 1e9,
                    peak=1024 * VALU_RATE_PER_SIMD / 1e9, unit="G wave-instructions/s", frac=exact["frac"],
                    kernel_ms=exact["kernel_ms_rocprofv3"], traffic=traffic,
                    hbm_frac_measured=None if traffic is None else
                    traffic / (exact["kernel_ms_rocprofv3"] * 1e-3) / 1e9 / HBM_PEAK_GBPS)
    ach = scan_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    return dict(common, bound="hbm", bound_measured="not measured in this run (no rocprofv3 child run): the contractual "
                                                    "algorithmic rate only", kernel="scan launches of a batch",
                achieved=ach, peak=HBM_PEAK_GBPS, unit="GB/s (algorithmic)", frac=None, traffic=traffic)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n", type=int, default=1183514)
    ap.add_argument("--d", type=int, default=100)
    ap.add_argument("--n-clusters", type=int, default=1087)
    ap.add_argument("--nq", type=int, default=10000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--n-probes", type=int, default=10)
    ap.add_argument("--seed", type=int, default=10)
    ap.add_argument("--fit-sample", type=int, default=100000)
    ap.add_argument("--cpu-sample", type=int, default=120000,
                    help="queries timed on the CPU oracle (one core, ~10 s at the default) and compared row by row: the "
                         "distinct batches of the timed region, then more batches of the same generator")
    ap.add_argument("--recall-sample", type=int, default=1000)
    ap.add_argument("--cache-dir", default=os.environ.get("TMPDIR", "/tmp"))
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--build-probes", type=int, default=1, help="lists per point (ivf.py:53)")
    ap.add_argument("--metric", choices=["angular", "euclidean"], default="angular")
    ap.add_argument("--data", choices=["glove-like", "sift-like", "sift-clustered"], default="glove-like",
                    help="synthetic stand-in: Gaussian clusters, or |N(0,1)|*40 clipped to [0,218]")
    ap.add_argument("--heap-mode", type=int, default=0, help="tk_index_set_heap_mode (A/B)")
    ap.add_argument("--rank-share-links", choices=["dense", "both"], default="dense",
                    help="rank-share leg: `both` also runs the filtered exchange for the links table (+1 minute; "
                         "profiles/r06/rank_share_links.txt)")
    ap.add_argument("--c5-micro", type=int, default=0,
                    help="--workload c5: every one of the 3000 clusters as this many micro-clusters (0: the stand-in of "
                         "rounds 2-5, isotropic noise around 3000 centres)")
    ap.add_argument("--c5-sigma", type=float, default=0.15, help="point noise around a micro-cluster's centre (--c5-micro)")
    ap.add_argument("--sort-queries", choices=["none", "probe", "probe-xcd"], default="none",
                    help="A/B: order every batch's queries by their nearest coarse centre before the upload")
    ap.add_argument("--scan-mode", type=int, default=0, help="tk_index_set_scan_mode (A/B)")
    ap.add_argument("--rescore-form", type=int, default=2, help="TK_OPT_RESCORE_FORM (A/B): 2 = 32-row tiles, 1 = 64-row, 0 = row per lane")
    ap.add_argument("--scan-form", type=int, default=0,
                    help="tk_index_set_option TK_OPT_SCAN_FORM (A/B): 0 table rows by per-lane global loads (default, fastest); "
                         "1 rows through LDS, 4 waves/SIMD; 2 LDS, 3 waves/SIMD")
    ap.add_argument("--replay-lazy", type=int, default=-1, choices=[-1, 0, 1],
                    help="TK_OPT_REPLAY_LAZY (A/B): -1 = by the index (lazy for long lists), 0 = staged, 1 = lazy")
    ap.add_argument("--pipeline", type=int, default=2,
                    help="replay streams (tk_index_set_pipeline): caller + coarse stream + these = 4 HW queues")
    ap.add_argument("--coalesce", type=int, default=2, choices=[1, 2],
                    help="tk_index_set_coalesce: 2 = pairs of consecutive steps run through the pipeline as one "
                         "batch of 2 x nq queries (same rows out; the latency-bound kernels cost the same for both)")
    ap.add_argument("--sweep", choices=["auto", "none"], default="auto",
                    help="auto: n_probes 1/5/20/50, build_probes=2 and the SIFT-shaped index beside the headline "
                         "(default workload at N = 1 only)")
    ap.add_argument("--workload", choices=["glove", "c5"], default="glove",
                    help="c5: BASELINE configs[4] on one GPU, 100M x 128 generated and built in HBM; "
                         "implies --n 100000000 --d 128 --n-clusters 10000 --metric euclidean unless given")
    ap.add_argument("--shard", choices=["auto", "none", "lists"], default="auto",
                    help="list-sharded leg after the replica measurement (auto: when N > 1)")
    ap.add_argument("--shard-depth", type=int, default=8,
                    help="list-sharded leg: batches in flight (each on its own stream)")
    ap.add_argument("--shard-coalesce", type=int, default=0,
                    help="list-sharded leg: consecutive steps answered as ONE sharded batch (<= 131072 queries); "
                         "0 = max(6, N)")
    ap.add_argument("--shard-exchange", choices=["auto", "dense", "filtered", "both"], default="auto",
                    help="list-sharded leg: whole distance segments at fixed positions (no host "
                         "synchronisation), or SURVEY 8e's filtered records; both: dense is reported, the "
                         "filtered run beside it; auto: filtered where the lists are long against the heap "
                         "(size-weighted mean list >= 32 heap sizes: the 100M workload), dense otherwise")
    ap.add_argument("--shard-coarse", choices=["home", "replicated"], default="home",
                    help="list-sharded leg: coarse stage of the home queries + probe all-gather, or of all "
                         "queries on every rank")
    ap.add_argument("--caller-stream", choices=["null", "own"], default="null",
                    help="stream of the timed region (A/B: the NULL stream, or a stream of its own)")
    ap.add_argument("--no-hbm-leg", action="store_true", help="skip the >= 1 GiB streaming scan leg")
    ap.add_argument("--traffic", choices=["auto", "none"], default="auto",
                    help="auto: HBM bytes of the scan launch from rocprofv3 --pmc child runs of this "
                         "script (N = 1, default workload only)")
    ap.add_argument("--py-cpu-sample", type=int, default=3000,
                    help="queries of the per-query Python-loop CPU baseline (examples/bench.py:118-137)")
    ap.add_argument("--profile-only", action="store_true",
                    help="stop after the timed region + the isolated stages (profiler runs)")
    ap.add_argument("--shard-counts", default="device", choices=["device", "host"],
                    help="filtered exchange: record counts read on the device (fixed regions, no host "
                         "synchronisation) or on the host (variable splits)")
    ap.add_argument("--shard-plain", type=int, default=1, choices=[0, 1, 2, 3],
                    help="list-sharded leg: 1 = the matrix-core kernel, one phase for the dense exchange / two phases for "
                         "the filtered one (default; the dense form falls back to 3 by itself where its check at home "
                         "fails); 2 = two phases for both (A/B); 3 = one phase behind the head bounds; 0 = exact kernel only")
    ap.add_argument("--rank-share", type=int, default=8,
                    help="N = 1: after the list-sharded leg, time ONE rank's share of a partition over this many ranks "
                         "(peers simulated on this GPU); 0 = skip")
    ap.add_argument("--shard-limit", type=float, default=240.0,
                    help="seconds after which a stuck list-sharded leg is abandoned")
    ap.add_argument("--force-collectives", action="store_true",
                    help="N = 1 with --shard lists: a one-rank nccl process group, the sharded leg's "
                         "exchanges through RCCL instead of device copies (rehearsal of the N > 1 code path)")
    ap.add_argument("--graph-steps", type=int, default=32, help="steps captured in the hipGraph of the pipelined mode")
    ap.add_argument("--batches", type=int, default=4,
                    help="distinct query batches rotated through the timed loop")
    ap.add_argument("--warmup-seconds", type=float, default=0.5,
                    help="warm up for at least this long, whatever --warmup says")
    ap.add_argument("--windows", type=int, default=0,
                    help="windows of --steps steps in the timed region (0: ~2400 steps in all, 3..61 windows); "
                         "ms_per_step is the median window")
    ap.add_argument("--data-file", default=None,
                    help=".npy of float vectors (GloVe-100, SIFT-1M ...): the protocol of examples/bench.py:67-70 "
                         "(seed 10, shuffle, last --nq rows are the queries) instead of the synthetic stand-in")
    ap.add_argument("--query-file", default=None, help=".npy of queries for --data-file (else its last --nq rows)")
    ap.add_argument("--a", type=float, default=1.0,
                    help="--data-file: n_clusters = int(a * sqrt(n)) unless --n-clusters is given (examples/bench.py:73)")
    args = ap.parse_args()
    if args.workload == "c5":
        dflt = ap.parse_args([])
        if args.n == dflt.n: args.n = 100_000_000
        if args.d == dflt.d: args.d = 128
        if args.n_clusters == dflt.n_clusters: args.n_clusters = 10000
        args.metric = "euclidean"
        args.cpu_sample = min(args.cpu_sample, 300)
        args.py_cpu_sample = min(args.py_cpu_sample, 50)     # (N > 1: every rank generates and builds the
        #  same index from the seed, then keeps the codes of the lists it owns: tk_index_shard_resident)

    # HIP maps every stream of the process onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that
    # share one run in order.  The pipelined index needs four (caller, front, two replay streams: r04: more
    # gain nothing for it); a list-sharded rank keeps --shard-depth batches in flight, each a chain of ~50
    # short dependent kernels on its own stream — eight queues: 0.131 -> 0.114 ms per step of one rank's
    # share of W = 8 (profiles/r05/rank_share_depth_queues.txt).  Read once, when HIP initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist
    ang = args.metric == "angular"
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev_id = local_rank % torch.cuda.device_count()     # (== local_rank on a real node)
    torch.cuda.set_device(dev_id)
    device = torch.device("cuda", dev_id)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)
    elif args.force_collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        dist.init_process_group(args.backend, rank=0, world_size=1,
                                **({"device_id": device} if args.backend == "nccl" else {}))

    from tinyknn_amd import _lib
    _lib.check(_lib.lib().tk_set_device(dev_id))

    # -- index: rank 0 builds (or loads) first so that the cache exists for the others
    if world > 1 and rank != 0:
        dist.barrier()
    real_queries = None
    if args.workload == "c5":
        ivf, cent = build_index_c5(args, device)
    elif args.data_file:
        X, real_queries = load_real(args)
        args.n, args.d = X.shape
        if args.n_clusters == ap.get_default("n_clusters"):
            args.n_clusters = int(args.a * args.n ** 0.5)          # examples/bench.py:73
        ivf, cent = build_index(args, device, X)
        del X
    else:
        ivf, cent = build_index(args, device)
    if world > 1 and rank == 0:
        dist.barrier()
    lap("imports, data, index fit + build (or cache load)")
    dev = ivf.device_index()
    lap("index upload (tk_index_set_lists, tk_index_set_data)")
    M = ivf.pq.centers.shape[1] // 2

    # -- this rank's batches, normalised on the host exactly like ivf.py:125-127.  The timed loop
    #    rotates N_BATCHES distinct batches (step i answers batch i % N_BATCHES into its own output
    #    rows): no step finds the previous step's probe lists, code lines or candidate rows warm
    #    because they were its own
    N_BATCHES = max(1, args.batches)

    def make_batch(b):
        if real_queries is not None:       # the file's queries, rotated by a quarter per batch
            return np.roll(real_queries, -b * (len(real_queries) // N_BATCHES), axis=0)[:args.nq].copy()
        if args.workload == "c5":
            return synth_rows_host(args.nq, args.d, args.seed + 100 + rank + 1000 * b, cent, getattr(args, "c5_point_sigma", 0.7))
        return synth_queries(cent, args.nq, args.seed + 100 + rank + 1000 * b, kind=args.data)

    batches = []
    for b in range(N_BATCHES):
        qs_b = make_batch(b)
        if args.sort_queries != "none":
            # A/B (round 6): the batch's queries ordered by the coarse centre nearest to them (the first probed list,
            # approximately) — neighbours in the batch then rescore overlapping candidate rows; "probe-xcd": ... and
            # dealt out so that workgroup b of a one-workgroup-per-query kernel (XCD b % 8) gets sorted position
            # (b % 8) * (nq / 8) + b / 8, i.e. each XCD's L2 sees one contiguous eighth of the sorted batch
            qn_tmp, _ = ivf._prepare(qs_b.copy())
            key = np.argmax(qn_tmp @ ivf.active_centers.T, axis=1) if ang else \
                np.argmin(((qn_tmp[:, None, :8] - ivf.active_centers[None, :, :8]) ** 2).sum(-1), axis=1)
            order = np.argsort(key, kind="stable")
            if args.sort_queries == "probe-xcd":
                nq8 = len(order) // 8
                pos = (np.arange(8 * nq8) % 8) * nq8 + np.arange(8 * nq8) // 8
                order = np.concatenate([order[:8 * nq8][pos], order[8 * nq8:]])
            qs_b = np.ascontiguousarray(qs_b[order])
        t_prep = time.perf_counter()
        qn_b, qp_b = ivf._prepare(qs_b.copy())
        t_prep = time.perf_counter() - t_prep
        batches.append(dict(qs=qs_b, qn=qn_b, qp=qp_b, q_dev=torch.from_numpy(qn_b).to(device),
                            qp_dev=torch.from_numpy(np.ascontiguousarray(qp_b)).to(device),
                            out=torch.full((args.nq, args.k), -1, dtype=torch.int64, device=device)))
    qs, qn, qp = batches[0]["qs"], batches[0]["qn"], batches[0]["qp"]       # batch 0: parity, recall, side legs
    q_dev, qp_dev, out_dev = batches[0]["q_dev"], batches[0]["qp_dev"], batches[0]["out"]
    qp_is_f64 = qp.dtype != np.float32      # rotated PQ: float64 table-build queries
    own_stream = None
    if args.caller_stream == "own":       # A/B: a non-NULL, non-blocking stream carries the scan chain
        own_stream = torch.cuda.Stream()
        torch.cuda.set_stream(own_stream)
    stream = torch.cuda.current_stream().cuda_stream
    dev.set_pipeline(args.pipeline)
    dev.set_coalesce(args.coalesce if args.pipeline > 1 else 1)
    # (labels that repeat — build_probes >= 2 — pair up where the index has its twin table: heap.hip, TWIN form)
    args.pairs_fit = (2 * args.nq <= dev.max_sub_batch(args.k, args.n_probes)
                      and (args.build_probes == 1 or dev.twin_table_width() > 0))
    dev.reserve(args.nq * launch_batches(args), args.k, args.n_probes)
    dev.set_heap_mode(args.heap_mode)
    dev.set_scan_mode(args.scan_mode)
    dev.set_option(_lib.OPT_SCAN_FORM, args.scan_form)
    dev.set_option(_lib.OPT_RESCORE_FORM, args.rescore_form)
    dev.set_option(_lib.OPT_REPLAY_LAZY, args.replay_lazy)
    n_step = [0]

    def step(done_event=None):
        b = batches[n_step[0] % N_BATCHES]
        n_step[0] += 1
        dev.query_batch_dev(b["q_dev"].data_ptr(), b["qp_dev"].data_ptr(), qp_is_f64, args.nq, args.k,
                            args.n_probes, b["out"].data_ptr(), stream=stream, done_event=done_event)

    # warm-up: --warmup steps, and at least --warmup-seconds of steps whatever the flag says (clocks,
    # caches, the allocator and every workspace of the pipeline settle; the driver's 5 steps are 4 ms)
    tw = time.perf_counter()
    n_warm = 0
    while n_warm < args.warmup or time.perf_counter() - tw < args.warmup_seconds:
        for _ in range(8):
            step()
        n_warm += 8
        dev.join(stream)
        torch.cuda.synchronize()
    n_step[0] = 0
    if world > 1:
        dist.barrier()
    dev.set_profiling(4)        # HIP events around the stages of every 4th batch, read after the region
    # Timed region: `windows` x --steps steps in one continuous run, bracketed by barrier +
    # synchronize on both sides.  A completion event (recorded by the library behind the last kernel
    # of a batch, on whichever internal stream it ran) closes every window of EXACTLY --steps steps;
    # ms_per_step = the median window.  Why not one window: with several batches in flight a region
    # pays the pipeline's fill and drain once (~1.5 ms: 10 % of a 20-step region, 1 % of a 200-step
    # one) — the windows behind the first measure what a server that keeps submitting sustains,
    # the whole-region figure (fill and drain included) is reported beside it as `drained`.
    K = args.steps
    # (round 6: ~2 400 steps — 61 windows of the driver's 20 — instead of ~600: the four streams fall into and out of a slower
    #  phase that lasts 40-100+ ms at a time; the median of half a second sees both where 0.12 s saw one)
    target_steps = 600 if args.workload == "c5" else 2400
    n_win = args.windows if args.windows > 0 else max(3, min(61, -(-target_steps // K)))
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_win + 1)]
    for e in evs:
        e.record()              # materialises the HIP event; the library re-records it
    torch.cuda.synchronize()
    # (a completion event belongs to ONE sub-batch of the index: a batch that the index cuts into
    #  several — distance rows beyond a workspace — closes its windows by draining instead)
    by_events = args.nq <= dev.max_sub_batch(args.k, args.n_probes)
    t0 = time.perf_counter()
    evs[0].record()
    for w in range(n_win):
        for i in range(K):
            step(evs[w + 1].cuda_event if (by_events and i == K - 1) else None)
        if not by_events:
            dev.join(stream)
            evs[w + 1].record()
    host_enqueue = time.perf_counter() - t0     # the host's share: every call of the region issued
    dev.join(stream)            # the caller's stream waits for every batch in flight
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    region = time.perf_counter() - t0
    win_ms = [evs[w].elapsed_time(evs[w + 1]) for w in range(n_win)]      # window 0 includes the fill
    steady = sorted(win_ms[1:]) if n_win > 1 else win_ms
    elapsed = steady[len(steady) // 2] * 1e-3        # seconds per --steps steps: the median window
    stages, scan_bytes, n_prof = dev.last_profile()
    plain_ms_timed = getattr(dev, "last_plain_kernel_ms", 0.0)      # HIP events around the plain kernel alone
    plain_stats_timed = dev.plain_stats()       # of the last batch of the timed region (a pair of steps when coalescing)
    raw_leg = None if args.profile_only else raw_stream_leg(args, dev, qs, out_dev, device, world)
    lap("queries, warm-up, timed region, raw stream leg")
    # the same kernels with ONE batch in flight (no co-running batches), for reference
    dev.set_pipeline(1)
    dev.reserve(args.nq, args.k, args.n_probes)
    step()
    torch.cuda.synchronize()     # untimed: first call after the workspace change
    dev.set_profiling(True)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    iso_stages, iso_bytes, _ = dev.last_profile()
    dev.set_profiling(False)
    plain_stats = dev.plain_stats()      # of the last batch: tiles, pairs by kernel, queries re-scanned
    # the list replay's insert rounds of ONE batch (roofline.replay): a device counter, one extra step
    replay_rounds = None
    try:
        dev.set_option(_lib.OPT_REPLAY_COUNT, 1)
        step()
        torch.cuda.synchronize()
        replay_rounds = dev.replay_stats()
        dev.set_option(_lib.OPT_REPLAY_COUNT, 0)
    except Exception as e:      # noqa: BLE001 - measurement plumbing
        log(f"[bench] replay rounds not counted: {e!r}")
    if args.profile_only:
        if rank == 0:
            print(json.dumps({"profile_only": True, "ms_per_step": elapsed / args.steps * 1e3,
                              "host_enqueue_ms_per_step": host_enqueue / (n_win * K) * 1e3,
                              "ms_per_step_drained": region / (n_win * K) * 1e3, "windows_ms": win_ms,
                              "stage_ms": stages, "isolated_stage_ms": iso_stages, "plain_kernel_ms": plain_ms_timed,
                              "scan_bytes": scan_bytes, "iso_scan_bytes": iso_bytes}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    # BASELINE configs[3]: a hipGraph-captured batch.  (a) one batch, one batch in flight, captured
    # once and replayed; (b) the PIPELINED mode as one graph: `gsteps` consecutive steps (the rotating
    # batches) + the join, captured on a side stream — the index's internal streams fork from it
    # through the events the calls record and are joined again by tk_index_join — and replayed.
    graph = None
    try:
        if world > 1:      # RCCL's watchdog thread touches the device during a global-mode capture
            raise RuntimeError("skipped at N > 1")
        g = torch.cuda.CUDAGraph()
        gout = torch.full((args.nq, args.k), -1, dtype=torch.int64, device=device)
        with torch.cuda.graph(g):
            dev.query_batch_dev(q_dev.data_ptr(), qp_dev.data_ptr(), qp_is_f64, args.nq, args.k,
                                args.n_probes, gout.data_ptr(),
                                stream=torch.cuda.current_stream().cuda_stream)
        g.replay()
        torch.cuda.synchronize()
        tg = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        tg = (time.perf_counter() - tg) / 20
        graph = {"one_batch_in_flight": {"ms_per_replay": tg * 1e3, "queries_per_s": args.nq / tg,
                                         "identical_to_stream_launch": bool((gout.cpu().numpy() == out_dev.cpu().numpy()).all())}}
        del g
        # (b)
        gsteps = args.graph_steps
        dev.set_profiling(False)
        dev.set_pipeline(args.pipeline)
        dev.reserve(args.nq, args.k, args.n_probes)
        gouts = [torch.full((args.nq, args.k), -1, dtype=torch.int64, device=device) for _ in range(N_BATCHES)]

        def gstep(i, st):
            bb = batches[i % N_BATCHES]
            dev.query_batch_dev(bb["q_dev"].data_ptr(), bb["qp_dev"].data_ptr(), qp_is_f64, args.nq, args.k,
                                args.n_probes, gouts[i % N_BATCHES].data_ptr(), stream=st)

        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for i in range(gsteps):      # uncaptured once: workspaces, events and streams exist afterwards
                gstep(i, side.cuda_stream)
            dev.join(side.cuda_stream)
        torch.cuda.synchronize()
        dev.quiesce()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            cs = torch.cuda.current_stream().cuda_stream
            for i in range(gsteps):
                gstep(i, cs)
            dev.join(cs)
        dev.quiesce()
        for o in gouts:
            o.fill_(-1)
        g2.replay()
        torch.cuda.synchronize()
        same = all(bool((gouts[b_].cpu().numpy() == batches[b_]["out"].cpu().numpy()).all()) for b_ in range(N_BATCHES))
        tg = time.perf_counter()
        for _ in range(10):
            g2.replay()
        torch.cuda.synchronize()
        tg = (time.perf_counter() - tg) / 10
        graph.update({"ms_per_replay": tg * 1e3, "steps_per_replay": gsteps, "ms_per_step": tg / gsteps * 1e3,
                      "queries_per_s": args.nq * gsteps / tg, "batches_in_flight": args.pipeline,
                      "identical_to_stream_launch": same,
                      "note": "the pipelined mode captured as ONE hipGraph (%d steps + join; internal streams "
                              "forked and joined by events inside the capture), replayed; each replay pays the "
                              "pipeline's fill and drain once" % gsteps})
        del g2
        dev.quiesce()
        dev.set_pipeline(1)
    except Exception as e:       # capture support is an extra, never the measured path
        graph = dict(graph or {}, error=repr(e))
        try:
            dev.quiesce()
            dev.set_pipeline(1)
        except Exception:
            pass
    # the same batch through the HOST-pointer entry point (tk_index_query_batch: H2D of the
    # queries, the pipeline, D2H of the ids, synchronous) — the PCIe-inclusive rate
    dev.query_batch(qn, qp, args.k, args.n_probes)
    th = time.perf_counter()
    for _ in range(3):
        host_ids = dev.query_batch(qn, qp, args.k, args.n_probes)
    host_qps = 3 * args.nq / (time.perf_counter() - th)
    # the same RAW queries through the device front end ("fast mode": normalisation, padding,
    # rotation on the GPU instead of numpy's per-query BLAS calls; within 1 ulp, not exact)
    try:
        fast_ids = dev.query_batch_raw(qs, args.k, args.n_probes)
        th = time.perf_counter()
        for _ in range(3):
            fast_ids = dev.query_batch_raw(qs, args.k, args.n_probes)
        fast_qps = 3 * args.nq / (time.perf_counter() - th)
    except (AssertionError, RuntimeError) as e:     # e.g. angular with d > 128: an extra only
        fast_ids, fast_qps = None, repr(e)
    t = torch.tensor([elapsed], dtype=torch.float64,
                     device=device if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    got = out_dev.cpu().numpy()
    got_all = [b_["out"].cpu().numpy() for b_ in batches]      # every distinct batch of the timed region

    do_shard = args.shard == "lists" or (args.shard == "auto" and world > 1)
    # N = 1, default workload: the list-sharded leg as a ONE-rank rehearsal, every exchange through RCCL
    # (torch.distributed, backend nccl, world 1), both exchanges, coalesced and fixed-Q — so that the
    # driver's line carries it.  The process group is created only now: RCCL's internal stream would share
    # one of HIP's four hardware queues with the pipelined index during the headline region.
    shard_w1 = (args.shard == "auto" and world == 1 and not args.data_file and not args.profile_only and
                (args.workload, args.n, args.d, args.n_clusters, args.nq, args.n_probes, args.metric, args.data,
                 args.build_probes) == ("glove", 1183514, 100, 1087, 10000, 10, "angular", "glove-like", 1))
    if rank != 0:
        if do_shard:
            import threading
            def give_up():
                log(f"[bench] rank {rank}: list-sharded leg stuck for {args.shard_limit + 120:.0f}s, exiting 3")
                os._exit(3)

            wd = threading.Timer(args.shard_limit + 120.0, give_up)
            wd.daemon = True
            wd.start()
            try:
                list_sharded_leg(args, ivf, cent, dev, device, world, rank)
            except Exception as e:      # rank 0 reports; a stuck collective ends by the timer
                log(f"[bench] rank {rank}: list-sharded leg failed: {e!r}")
            wd.cancel()
        if world > 1:
            dist.destroy_process_group()
        return

    qps = args.nq * world * args.steps / elapsed
    scan_ms = stages["scan"]
    achieved = scan_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    traffic, traffic_src = None, "not measured"
    default_wl = (args.workload, args.n, args.d, args.n_clusters, args.nq, args.n_probes, args.metric,
                  args.data, args.build_probes) == ("glove", 1183514, 100, 1087, 10000, 10, "angular",
                                                    "glove-like", 1)
    kstats, kstats_src = None, "not run"
    plain_on_timed = bool(plain_stats_timed and plain_stats_timed.get("plain_units"))
    if args.traffic == "auto" and (default_wl or args.workload == "c5") and world == 1 and not args.data_file:
        lap("hipgraph leg, stage and isolated timings")
        traffic, traffic_src = measure_traffic(args, plain_on_timed)
        lap("PMC traffic child runs")
        kstats, kstats_src = kernel_stats_child(args, plain_on_timed)
        lap("kernel-stats child run")
    hbm_leg = None
    if not args.no_hbm_leg and world == 1:
        try:
            hbm_leg = hbm_scale_leg(device)
        except Exception as e:      # noqa: BLE001 - an extra leg must not lose the line
            hbm_leg = {"error": repr(e)}

    # the box's device-copy bandwidth (read + write), the practical HBM ceiling next to the spec
    try:
        src = torch.empty(1 << 28, dtype=torch.float32, device=device)      # 1 GiB
        dst = torch.empty_like(src)
        dst.copy_(src)
        torch.cuda.synchronize()
        tc0 = time.perf_counter()
        for _ in range(5):
            dst.copy_(src)
        torch.cuda.synchronize()
        copy_gbps = 5 * 2 * src.numel() * 4 / (time.perf_counter() - tc0) / 1e9
        # ... and a read-only kernel over 1 GiB (the flat scan's access pattern): the streaming-read
        # ceiling the HBM-scale scan leg is up against (profiles/r02_read_bw_microbench.txt)
        del src, dst
        import ctypes
        rb = ctypes.c_double(0.0)
        _lib.check(_lib.lib().tk_measure_read_bandwidth(1 << 30, 10, ctypes.byref(rb)))
        read_gbps = rb.value
    except Exception:
        copy_gbps = read_gbps = None
    if hbm_leg and "GBps" in hbm_leg and read_gbps:
        hbm_leg["frac_of_measured_read_only_kernel"] = hbm_leg["GBps"] / read_gbps

    # -- recall against brute force (torch matmul on the GPU: measurement plumbing)
    rs = min(args.recall_sample, args.nq)
    truth_s = None
    truth = None
    if ivf.data.dtype == np.float32 and args.d <= 128:
        # exact ground truth on the f32 matrix cores (brute.hip: numpy's knn_brute distances
        # bit for bit); qn and IVF.data are normalised for the angular metric
        try:
            tg = time.perf_counter()
            truth = dev.knn_brute(qn[:rs], args.k)
            truth_s = time.perf_counter() - tg
            recall = float(np.mean([len(set(truth[i]) & set(got[i])) / args.k for i in range(rs)]))
        except Exception as e:      # noqa: BLE001 - e.g. a candidate list overflow: fall back below
            log(f"[bench] tk_index_knn_brute failed ({e!r}); recall by torch matmul + topk")
            truth = None
    if truth is None and isinstance(ivf.data, np.ndarray):
        data_t = torch.from_numpy(ivf.data).to(device)
        sims = q_dev[:rs] @ data_t.T
        if not ang:   # squared euclidean: smallest |x|^2 - 2 q.x
            sims = 2 * sims - (data_t * data_t).sum(1)[None]
        truth = sims.topk(args.k, dim=1).indices.cpu().numpy()
        recall = float(np.mean([len(set(truth[i]) & set(got[i])) / args.k for i in range(rs)]))
        del data_t, sims
    elif truth is None:
        recall = None

    # -- CPU baseline: the oracle (a C port of the reference path), one thread, a
    #    bounded sample of the same batch; also a full-size parity check of the ids
    cpu = None
    parity = None
    query1 = None
    if not args.no_cpu and world == 1:
        cs = min(args.cpu_sample, args.nq)
        heap_parity = None
        if args.workload == "c5":
            ox, fill_rows, ox_cleanup = oracle_index_resident(ivf, args.cache_dir)
            # the rows the oracle will rescore = the heap candidates; the device's heaps are
            # compared with the oracle's first (probe lists and heap arrays, layout included)
            dev.set_pipeline(1)
            _, dbg = dev.query_batch(qn[:cs], qp[:cs], args.k, args.n_probes, debug=True)
            fill_rows(dbg["heap_idx"])
            same_h = same_p = 0
            hs = min(cs, 40)
            for i in range(hs):
                _, w = ox.query(qn[i], args.k, args.n_probes, debug=True)
                same_p += int(np.array_equal(w["probes"], dbg["probes"][i]))
                same_h += int(np.array_equal(w["heap_idx"], dbg["heap_idx"][i]) and
                              np.array_equal(w["heap_val"], dbg["heap_val"][i]))
            heap_parity = {"queries": hs, "identical_probe_lists": same_p, "identical_heap_arrays": same_h}
        else:
            ox = oracle_index(ivf)
        # The sample: the distinct batches of the timed region (their rows as the timed region left them), then further
        # batches of the same generator answered by the same pipelined calls — --cpu-sample queries in all (default
        # 120 000: ~10 s of one host core), EVERY row compared.
        cs_all = cs if args.workload == "c5" else max(1, args.cpu_sample)
        n_b = (cs_all + args.nq - 1) // args.nq
        sample_q, sample_got = [], []
        extra = []
        for b_ in range(n_b):
            m = min(args.nq, cs_all - b_ * args.nq)
            if b_ < N_BATCHES:
                sample_q.append(batches[b_]["qn"][:m])
                sample_got.append((got if b_ == 0 else got_all[b_])[:m])
            else:
                qn_x, qp_x = ivf._prepare(make_batch(b_).copy())
                extra.append(dict(m=m, qn=qn_x, q_dev=torch.from_numpy(qn_x).to(device),
                                  qp_dev=torch.from_numpy(np.ascontiguousarray(qp_x)).to(device),
                                  out=torch.full((args.nq, args.k), -1, dtype=torch.int64, device=device)))
        if extra:
            dev.set_pipeline(args.pipeline)
            dev.set_coalesce(args.coalesce if args.pipeline > 1 else 1)
            for x in extra:
                dev.query_batch_dev(x["q_dev"].data_ptr(), x["qp_dev"].data_ptr(), qp_is_f64, args.nq, args.k,
                                    args.n_probes, x["out"].data_ptr(), stream=stream)
            dev.join(stream)
            torch.cuda.synchronize()
            for x in extra:
                sample_q.append(x["qn"][:x["m"]])
                sample_got.append(x["out"].cpu().numpy()[:x["m"]])
            del extra
        tcpu, same_rows, want = 0.0, 0, None
        for q_b, g_b in zip(sample_q, sample_got):
            tc = time.perf_counter()
            want_b = ox.query_batch(q_b, args.k, args.n_probes)
            tcpu += time.perf_counter() - tc
            same_rows += int((want_b == g_b).all(axis=1).sum())
            if want is None:
                want = want_b
        cs = sum(len(q_b) for q_b in sample_q)
        cpu = {"value": cs / tcpu, "unit": "queries/s", "cores": 1, "kind": "port",
               "sample": (f"first {cs} queries of the same batch" if n_b == 1 else
                          f"{cs} queries = {n_b} batches of the same workload (the {min(n_b, N_BATCHES)} distinct batches of "
                          f"the timed region{'' if n_b <= N_BATCHES else ' + %d more of the same generator' % (n_b - N_BATCHES)})") +
                         f", oracle/tinyknn_oracle.c (AVX2 pshufb scan + sequential heap), {tcpu:.2f}s"}
        parity = {"queries_checked": cs, "identical_rows": same_rows}
        if heap_parity:
            parity.update(heap_parity)
        # like-for-like with examples/bench.py:118-137: ONE Python-level query() per query
        # (normalise, table, coarse top, chained list scans, rescoring per call), one thread
        ps = min(args.py_cpu_sample, args.nq)
        tp = time.perf_counter()
        for i in range(ps):
            q = np.ascontiguousarray(qs[i], dtype=np.float32).copy()      # ivf.py:125-127
            if ang:
                q /= np.linalg.norm(q)
            ox.query(q, args.k, args.n_probes)
        tpy = time.perf_counter() - tp
        cpu["python_loop"] = {"value": ps / tpy, "unit": "queries/s", "cores": 1,
                              "sample": f"first {ps} raw queries, one Python-level query() per query as "
                                        f"examples/bench.py:118-137 times the reference; the per-query work is the C port's "
                                        f"(table build and rescoring in C, where the reference runs numpy), {tpy:.2f}s"}
        # ... and the SAME protocol on the drop-in call of this library: `ivf.query(q)` per query (host preparation,
        # H2D, the device pipeline with the wave-per-query register heap, D2H, one call at a time), checked row by row
        # against the oracle's answers for the same queries
        if args.workload != "c5" and hasattr(ivf, "query"):
            try:
                dev.set_pipeline(1)
                nq1 = min(1000, args.nq)
                for i in range(20):
                    ivf.query(qs[i].copy(), args.k, n_probes=args.n_probes)
                t1 = time.perf_counter()
                got1 = [ivf.query(qs[i].copy(), args.k, n_probes=args.n_probes) for i in range(nq1)]
                t_q1 = (time.perf_counter() - t1) / nq1
                same1 = 0
                for i in range(nq1):
                    w1 = want[i] if i < len(want) else ox.query(qn[i], args.k, args.n_probes)
                    w1 = np.asarray(w1)
                    g1 = np.asarray(got1[i])
                    w1 = w1[w1 != -1] if len(g1) < args.k else w1
                    same1 += int(len(g1) == len(w1) and (g1 == w1).all())
                query1 = {"ms_per_query": t_q1 * 1e3, "queries_per_s": 1.0 / t_q1, "rows": nq1, "identical_rows": same1,
                          "cpu_oracle_ms_per_query": tpy / ps * 1e3,
                          "protocol": "examples/bench.py:118-137: one ivf.query(q) per query in a Python loop, raw "
                                      "float32 vector in, ids out, nothing in flight between calls"}
            except Exception as e:      # noqa: BLE001 - an extra leg must not lose the line
                query1 = {"error": repr(e)}
        if args.workload == "c5":
            ox_cleanup()
        cpu["note"] = ("value = C batch loop of the port (no per-query Python overhead: the STRONGER "
                       "baseline); python_loop = the reference's own measurement protocol")

    lap("hbm-scale leg, recall, CPU baseline, query1")
    # -- the two biggest consumers of GPU time beside the scan, each against the resource that binds it
    side_roofs = {}
    try:
        R_heap = (args.n_probes + 1) * args.k + 1
        if replay_rounds and replay_rounds["waves"]:
            mx, waves = replay_rounds["max_rounds_of_a_wave"], replay_rounds["waves"]
            iso_ms, timed_ms = iso_stages["heap"], stages["heap"]
            # a round = one insert step of a wave: every lane with a pending candidate sifts it down its own heap
            # (log2(R) levels: 3 in registers, the rest one dependent LDS round trip each) + the block bookkeeping
            # around it; profiles/r02_replay_phases.md counted ~240 dependent VALU (4-5 cycles each, one wave on its
            # SIMD) + 4 LDS round trips (~128 cycles each) per round at R = 111: ~1 500 cycles = the floor used here
            floor_cyc = 1500.0
            clock_ghz = 2.1
            floor_ms = mx * floor_cyc / (clock_ghz * 1e6)
            side_roofs["replay"] = {
                "kernel": "heap_replay_lanes_kernel (the probed lists through one heap per query, lane per query)",
                "bound": "latency (a dependent chain per wave; %d waves on 1024 SIMDs: no pipe is busy)" % waves,
                "insert_rounds_of_the_slowest_wave": mx, "insert_rounds_mean_per_wave": replay_rounds["rounds"] / waves,
                "waves": waves, "segments_walked_per_wave": replay_rounds["segments"] / waves,
                "kernel_ms_isolated": iso_ms, "stage_ms_timed_region": timed_ms,
                "ns_per_round_isolated": iso_ms * 1e6 / max(mx, 1),
                "floor_ms": floor_ms, "floor_is": "rounds of the slowest wave x ~1 500 cycles per round at 2.1 GHz (the chain "
                                                  "of one insert step: ~240 dependent VALU + 4 LDS round trips, "
                                                  "profiles/r02_replay_phases.md); the stage also holds pad_fix and, for "
                                                  "flagged / wrapped queries, the wave-per-query replays",
                "frac": floor_ms / iso_ms if iso_ms > 0 else None,
                "frac_timed_region": floor_ms / timed_ms if timed_ms > 0 else None,
                "covers": "ONE batch of %d queries, one batch in flight (a pair of calls runs as one launch of twice the "
                          "waves in about the same time)" % args.nq}
        row_bytes = args.d * 4
        if row_bytes % 16 == 0 and row_bytes <= 1024 and ivf.data.dtype == np.float32:
            gb = ctypes_double()
            _lib.check(_lib.lib().tk_measure_gather_bandwidth(int(min(args.n, 4_000_000)) * row_bytes, row_bytes,
                                                              int(args.nq) * R_heap, 20, gb.ref))
            gathered = float(args.nq) * R_heap * row_bytes
            iso_ms, timed_ms = iso_stages["rescore"], stages["rescore"]
            side_roofs["rescore"] = {
                "kernel": "rescore_staged_kernel<32> (exact distances of the heap's candidates, top-k)",
                "bound": "hbm (random rows of %d bytes: %d per query)" % (row_bytes, R_heap),
                "gathered_bytes_per_step": gathered, "kernel_ms_isolated": iso_ms, "stage_ms_timed_region": timed_ms,
                "achieved": gathered / (iso_ms * 1e-3) / 1e9 if iso_ms > 0 else None, "unit": "GB/s",
                "peak": gb.value, "peak_is": "a kernel that only gathers as many random %d-byte rows out of a table of the same "
                                             "row size with the same access pattern (tk_measure_gather_bandwidth), this box; "
                                             "MI355X_MICROARCH.md: 5.5-5.6 TB/s for 1 152-byte rows, 8 TB/s streaming" % row_bytes,
                "frac": (gathered / (iso_ms * 1e-3) / 1e9) / gb.value if iso_ms > 0 and gb.value > 0 else None,
                "frac_timed_region": (gathered / (timed_ms * 1e-3) / 1e9) / gb.value if timed_ms > 0 and gb.value > 0 else None,
                "frac_of_hbm_peak": gathered / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if iso_ms > 0 else None}
    except Exception as e:      # noqa: BLE001 - extra entries must not lose the line
        side_roofs["error"] = repr(e)

    sweep = None
    if args.sweep == "auto" and default_wl and world == 1 and not args.data_file:
        try:
            sweep = sweep_leg(args, ivf, dev, batches, qp_is_f64, stream, device, cent)
        except Exception as e:      # noqa: BLE001 - an extra leg must not lose the line
            sweep = {"error": repr(e)}
    line = {
        "metric": f"queries/sec at Recall10@10 on GloVe-100 angular (synthetic stand-in), IVF+4-bit PQ, build_probes={args.build_probes}"
                  if (args.workload, args.data, args.metric, args.d) == ("glove", "glove-like", "angular", 100) else
                  f"queries/sec at Recall10@10, synthetic {args.n} x {args.d} float32 (BASELINE configs[4] on one GPU), "
                  f"IVF n_clusters={args.n_clusters} + 4-bit PQ"
                  if args.workload == "c5" else
                  f"queries/sec at Recall10@10, {args.data} {args.metric} d={args.d} (synthetic), IVF+4-bit PQ",
        "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "timing": {"windows": n_win, "steps_per_window": K, "window_ms": win_ms,
                   "ms_per_step_is": "median over the windows behind the first, each EXACTLY --steps steps of one "
                                     "continuous run (completion events of the batches; fill and drain of the "
                                     "pipeline are in `drained`)",
                   "drained": {"steps": n_win * K, "ms_per_step": region / (n_win * K) * 1e3,
                               "queries_per_s": args.nq * world * n_win * K / region,
                               "note": "the whole region by the host clock, barrier + synchronize on both "
                                       "sides, pipeline fill and drain included"},
                   "host_enqueue_ms_per_step": host_enqueue / (n_win * K) * 1e3,
                   "warmup_steps_run": n_warm, "distinct_batches": N_BATCHES},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int8 (saturating PQ sums) + f32 (tables, rescoring)",
        "data": "synthetic" if not args.data_file else f"file {os.path.basename(args.data_file)}",
        "config": {"workload": ("c5 (BASELINE configs[4] on one GPU): " +
                                (f"3000 x {args.c5_micro} micro-clusters (centre + 0.7 N(0,1)), point noise sigma {args.c5_sigma}, "
                                 if args.workload == "c5" and args.c5_micro > 0 else "") +
                                "3000 Gaussian clusters sigma 0.7 generated in "
                                "HBM (seeded counter-based generator), index built on the device, PQ rotated to 64 "
                                "dims, euclidean, "
                                if args.workload == "c5" else
                                f"{args.data} {args.metric} stand-in "
                                "(glove-like: 300 Gaussian clusters sigma 0.7; sift-like: |N(0,1)|*40 clipped): "
                                if (args.data, args.metric) != ("glove-like", "angular") else
                                "glove-100-angular stand-in: 300 Gaussian clusters sigma 0.7, ") +
                               f"N={args.n} d={args.d} IVF n_clusters={args.n_clusters} "
                               f"build_probes={args.build_probes} FastPQ dpb=2 M={M}",
                   "queries_per_step_per_gpu": args.nq, "k": args.k, "n_probes": args.n_probes,
                   "pass_1": (args.n_probes + 1) * args.k + 1, "recall10@10": recall,
                   "recall_queries": rs,
                   "recall_ground_truth": (None if truth_s is None else
                                           f"tk_index_knn_brute (f32 MFMA, exact): {rs} queries x {args.n} vectors "
                                           f"in {truth_s * 1e3:.0f} ms"),
                   "parallelism": f"replica x{world} (queries sharded)",
                   "batches_in_flight": args.pipeline},
        "roofline": roofline_entry(args, M, plain_stats_timed, plain_ms_timed, stages, iso_stages, scan_bytes, n_prof,
                                   traffic, traffic_src, kstats, kstats_src, copy_gbps, read_gbps),
        "raw_in_ids_out": raw_leg,
        "roofline_hbm_scale": hbm_leg,
        "stage_ms": stages,
        "isolated": {"note": "same batch with one batch in flight (5 steps after the timed region)",
                     "stage_ms": iso_stages, "ms_per_step": sum(iso_stages.values()),
                     "scan_stage_algorithmic_GBps": iso_bytes / (iso_stages["scan"] * 1e-3) / 1e9,
                     "scan_stage_note": "algorithmic bytes (one code byte per (query, code) pair) over the isolated scan "
                                        "stage; NOT HBM traffic (32 queries share each fetched tile, the code set sits in "
                                        "L2 / Infinity Cache): no fraction of the HBM peak is formed from it"},
        "hipgraph": graph,
        "host_boundary": {"queries_per_s": host_qps,
                          "note": "tk_index_query_batch with host buffers: H2D queries + pipeline + D2H ids, "
                                  "synchronous, one batch at a time (never `value`)",
                          "identical_to_device_path": bool((host_ids == got).all()),
                          "host_prepare_ms_per_batch": t_prep * 1e3,
                          "host_prepare_note": "ivf.py:125-128 + fast_pq.py:200-204 per query in numpy "
                                               "(BLAS norm, padding, BLAS GEMV), before any of the rates above"},
        "fast_front_end": {"queries_per_s": fast_qps,
                           "note": "raw float32 queries in, ids out (H2D + device normalisation/padding/rotation "
                                   "+ pipeline + D2H, synchronous); within 1 ulp of the host preparation, not "
                                   "bit-identical: never `value`",
                           "rows_identical_to_exact_path": (None if fast_ids is None else
                                                            int((fast_ids == got).all(axis=1).sum())),
                           "rows": args.nq},
        "cpu_baseline": cpu,
        "parity_vs_oracle": parity,
        "query1": query1,
        "sweep": sweep,
    }
    if isinstance(query1, dict) and "ms_per_query" in query1:
        line["query1_ms_per_query"] = query1["ms_per_query"]      # (scalar: the reference's own protocol on the drop-in call)
    # the other two chains of a batch, next to the scan kernel's entry (and their fractions as top-level scalars)
    if isinstance(line.get("roofline"), dict):
        line["roofline"]["replay"] = side_roofs.get("replay")
        line["roofline"]["rescore"] = side_roofs.get("rescore")
        if side_roofs.get("error"):
            line["roofline"]["replay_rescore_error"] = side_roofs["error"]
        for key in ("replay", "rescore"):
            if side_roofs.get(key) and side_roofs[key].get("frac") is not None:
                line["roofline_%s_frac" % key] = side_roofs[key]["frac"]
    if args.workload == "c5" and world == 1 and args.rank_share > 1 and not do_shard:
        # configs[4] is the workload north_star shards: one rank's share of a W-rank partition of THIS index
        # (the resident index stays unsharded: the simulated ranks are clone shards that borrow its arrays)
        try:
            dev.set_pipeline(1)
            qn_s, qp_s, want_s = shard_inputs(args, ivf, cent, dev, device)
            rs = rank_share_leg(args, ivf, device, qn_s, qp_s, want_s, args.rank_share)
            W = rs["world"]
            rs["unsharded_ms_per_step_over_W"] = line["ms_per_step"] / W
            rs["target_ms_per_step_at_0.7_efficiency"] = line["ms_per_step"] / (0.7 * W)
            rs["implied_strong_scaling_efficiency_without_links"] = line["ms_per_step"] / W / rs["ms_per_step"]
            _links_at_target(rs)
            line["rank_share_W%d" % W] = rs
            line["rank_share_W%d_ms_per_step" % W] = rs["ms_per_step"]
            line["rank_share_W%d_implied_efficiency_without_links" % W] = rs["implied_strong_scaling_efficiency_without_links"]
        except Exception as e:      # noqa: BLE001 - an extra leg must not lose the line
            line["rank_share_W%d" % args.rank_share] = {"error": repr(e)}
    lap("rank-share leg")
    if isinstance(raw_leg, dict) and "queries_per_s" in raw_leg:
        line["raw_in_ids_out_queries_per_s"] = raw_leg["queries_per_s"]      # (scalar: kept by the driver's parsed record)
    if shard_w1 and not do_shard:
        try:
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
            args.force_collectives = True
            args.shard_exchange = "both"
            args.backend = "nccl"
            do_shard = True
        except Exception as e:      # noqa: BLE001 - the rehearsal is an extra
            line["list_sharded"] = {"error": "one-rank process group: " + repr(e)}
    if do_shard:
        # the replica line above is complete: a list-sharded leg that gets stuck (it is the
        # one part that cannot be rehearsed on a 1-GPU box with RCCL) must not lose it
        import threading

        def bail():
            line["list_sharded"] = {"error": f"abandoned after {args.shard_limit:.0f}s"}
            emit(line)
            # a GPU process that abandoned a collective never reports success (N = 1: the headline stands)
            os._exit(3 if world > 1 else 0)

        wd = threading.Timer(args.shard_limit, bail)
        wd.daemon = True
        wd.start()
        try:
            line["list_sharded"] = ls = list_sharded_leg(args, ivf, cent, dev, device, world, rank)
        except Exception as e:
            # the other ranks may be waiting in a collective: do not join them again
            line["list_sharded"] = {"error": repr(e)}
            emit(line)
            os._exit(3 if world > 1 else 0)
        wd.cancel()
        if world == 1:
            ls["ratio_to_unsharded_value"] = ls["queries_per_s"] / line["value"]
            ls["rehearsal"] = ("ONE rank: every exchange goes through RCCL (all-to-all, all-gather, and for the "
                               "filtered exchange a uint8 MIN all-reduce, at world 1); what N ranks add is the links, "
                               "not the code path")
            # scalars at the top level: the driver's parsed record keeps those
            line["list_sharded_ratio_to_unsharded_value"] = ls["ratio_to_unsharded_value"]
            for key, rs in ls.items():
                if key.startswith("rank_share_W") and isinstance(rs, dict) and "ms_per_step" in rs:
                    W = rs["world"]
                    rs["unsharded_ms_per_step_over_W"] = line["ms_per_step"] / W
                    rs["target_ms_per_step_at_0.7_efficiency"] = line["ms_per_step"] / (0.7 * W)
                    rs["implied_strong_scaling_efficiency_without_links"] = line["ms_per_step"] / W / rs["ms_per_step"]
                    _links_at_target(rs)
                    line[key + "_ms_per_step"] = rs["ms_per_step"]
                    line[key + "_implied_efficiency_without_links"] = rs["implied_strong_scaling_efficiency_without_links"]
        if world > 1:
            # N > 1: the north_star split is the measured one; the replica rate stays beside it
            line["replica"] = {"queries_per_s": line["value"], "ms_per_step": line["ms_per_step"],
                               "scaling": "weak (every rank its own batch of %d queries, whole index per rank, "
                                          "no data-path collective)" % args.nq}
            line["value"] = ls["queries_per_s"]
            line["ms_per_step"] = ls["ms_per_step"]
            line["scaling"] = "strong"
            line["config"]["parallelism"] = (f"inverted lists sharded by cluster id x{world}, queries broadcast "
                                             f"(one shared batch of {args.nq} per step), RCCL all-gather of probe "
                                             "lists + all-to-all of int8 distances + all-gather of ids")
            line["config"]["queries_per_step_total"] = args.nq
            line["config"]["batches_in_flight"] = args.shard_depth
            for key in ("roofline", "stage_ms", "isolated", "hipgraph", "host_boundary", "fast_front_end",
                        "raw_in_ids_out"):
                if key in line:      # measured in the replica region: say so
                    line["replica"][key] = line.pop(key)
            line["roofline"] = dict(line["replica"]["roofline"],
                                    note="scan launch of the REPLICA region (the sharded leg runs the same "
                                         "kernel on this rank's 1/%d of the (query, list) segments)" % world)
    lap("list-sharded legs")
    emit(line)
    if world > 1 or dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
