"""The bench's FINAL stdout line stays small enough for a bounded tail of stdout to hold it whole.

Round 5's single line had grown to 20.9 KB and the driver's record lost `metric`, `value`, `config` and `roofline`
(`BENCH_r05.json parsed: null`).  `bench.emit` now prints the detail first (`# bench_detail …`, and a file) and a
compact line last; this test builds that line from a canned full result — round 5's own 20.9 KB record, kept under
`profiles/r05/`, and a deliberately bloated variant of it — and checks size, keys and JSON strictness.  No GPU.
"""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

CANNED = os.path.join(ROOT, "profiles", "r05", "bench_default.json")

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity_vs_oracle")


def canned():
    with open(CANNED) as f:
        return json.load(f)


def check(c, full):
    s = json.dumps(c, allow_nan=False)
    assert len(s) <= 4096, len(s)
    for k in CONTRACT_KEYS:
        assert k in c, k
    assert c["value"] == pytest.approx(full["value"], rel=1e-6)
    assert c["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-6)
    assert "workload" in c["config"] and "model" not in c["config"]
    for k in ("k", "n_probes", "recall10@10", "queries_per_step_per_gpu"):
        assert k in c["config"], k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in c["roofline"], k
    assert c["roofline"]["frac"] == pytest.approx(c["roofline"]["achieved"] / c["roofline"]["peak"], rel=1e-4)
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c["cpu_baseline"], k
    return s


def test_compact_line_of_round5_record():
    full = canned()
    assert len(json.dumps(full)) > 16000          # the record that did not parse
    c = bench.compact_line(full, "gpurun_out/bench/bench_detail.json")
    check(c, full)
    assert c["hipgraph_ratio"] == pytest.approx(full["hipgraph"]["queries_per_s"] / full["value"], rel=1e-3)
    assert c["sweep_b2_queries_per_s"] == pytest.approx(15757896.9, rel=1e-4)
    assert c["rank_share_W8_implied_efficiency_without_links"] == pytest.approx(
        full["rank_share_W8_implied_efficiency_without_links"], rel=1e-5)
    assert c["raw_in_ids_out_queries_per_s"] == pytest.approx(full["raw_in_ids_out_queries_per_s"], rel=1e-5)
    assert c["roofline"]["replay"]["frac"] == pytest.approx(full["roofline_replay_frac"], rel=1e-3)


def test_compact_line_stays_small_whatever_the_legs_return():
    full = canned()
    full["sweep"]["points"] = full["sweep"]["points"] * 40          # a sweep of 240 points
    full["metric"] = full["metric"] * 30
    full["config"]["workload"] = full["config"]["workload"] * 30
    full["cpu_baseline"]["sample"] = "x" * 5000
    full["roofline"]["kernel"] = "k" * 3000
    full["list_sharded"] = {"error": "e" * 10000}
    full["value"] = float(full["value"])
    full["roofline"]["hbm_frac_measured"] = float("nan")            # strict JSON: no NaN in the final line
    c = bench.compact_line(full, "d" * 64)
    check(c, full)
    assert c["roofline"]["hbm_frac_measured"] is None


def test_compact_line_without_optional_legs():
    full = canned()
    for k in ("sweep", "list_sharded", "hipgraph", "raw_in_ids_out", "raw_in_ids_out_queries_per_s", "cpu_baseline"):
        full.pop(k, None)
    full["cpu_baseline"] = None
    c = bench.compact_line(full)
    assert c["cpu_baseline"] is None and "sweep_points" not in c and len(json.dumps(c, allow_nan=False)) <= 4096


def test_emit_prints_the_detail_first_and_the_compact_line_last(tmp_path):
    full = canned()
    full["extra_nan"] = float("nan")
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(full, str(tmp_path / "detail.json"))
    lines = buf.getvalue().splitlines()
    assert len(lines) == 2
    assert lines[0].startswith("# bench_detail ")
    detail = json.loads(lines[0][len("# bench_detail "):])
    assert detail["sweep"] == full["sweep"] and detail["extra_nan"] is None
    last = json.loads(lines[1])
    check(last, full)
    with open(tmp_path / "detail.json") as f:
        assert json.load(f)["list_sharded"]["queries_per_s"] == full["list_sharded"]["queries_per_s"]


def test_bench_caches_round_trip(tmp_path):
    """The index / rows caches of bench.py: written by rename (no half-written file is ever visible under the final
    name), rows read back equal to synth()'s, a damaged or mis-shaped file regenerated instead of trusted."""
    import argparse
    import numpy as np
    import bench
    a = argparse.Namespace(cache_dir=str(tmp_path), n=3000, d=20, seed=10, data="glove-like")
    X1, c1 = bench.synth_cached(a)
    files = sorted(p.name for p in tmp_path.iterdir())
    assert files == ["tinyknn_bench_rows_n3000_d20_s10_glove-like.npy"], files      # (no .tmp left behind)
    X2, c2 = bench.synth_cached(a)
    X3, c3 = bench.synth(a.n, 0, a.d, a.seed, kind=a.data)
    assert np.array_equal(X1, X3) and np.array_equal(X2, X3) and np.array_equal(c1, c3) and np.array_equal(c2, c3)
    # a file of another shape under the same name (a stale cache) is not trusted
    np.save(tmp_path / files[0], np.zeros((5, 5), dtype=np.float32))
    X4, _ = bench.synth_cached(a)
    assert np.array_equal(X4, X3)
    # a truncated file neither
    (tmp_path / files[0]).write_bytes(b"\x93NUMPY\x01\x00garbage")
    X5, _ = bench.synth_cached(a)
    assert np.array_equal(X5, X3)
    # save_atomic: a failing writer leaves neither the final name nor the private one
    def boom(f):
        f.write(b"half")
        raise OSError("disk full")
    bench.save_atomic(str(tmp_path / "x.bin"), boom)
    assert not (tmp_path / "x.bin").exists() and not list(tmp_path.glob("x.bin.*"))
