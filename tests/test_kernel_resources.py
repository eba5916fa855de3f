"""Register / scratch budget of the hot kernels, read from the compiler (no GPU needed).

Round 3 shipped `scan_units2_kernel` with 480 B/lane of scratch: selecting among three by-value
job descriptors made hipcc copy them to private memory.  This test compiles the kernel files for
gfx950 with -Rpass-analysis=kernel-resource-usage and pins, for the instantiations the default
GloVe-shaped batch launches, zero scratch and the occupancy the design counts on.
"""
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tinyknn_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", "-o", os.devnull]

# (file, substring of the mangled name) -> (max VGPRs, min waves/SIMD)
PINNED = {
    # the fused exact launch of the pipelined mode: AVX order, signed tables, per-lane table loads
    ("adc_scan.hip", "scan_units2_kernelILi1ELb1ELi0EE"): (168, 3),
    ("adc_scan.hip", "scan_units_kernelILi1ELb1ELi0ELb0EE"): (168, 3),
    # the plain-sum scan on the matrix cores, M = 52 and M = 32
    ("plain_scan.hip", "scan_plain_wave_kernelILi26ELb1E"): (256, 2),
    ("plain_scan.hip", "scan_plain_wave_kernelILi16ELb1E"): (256, 2),
    # lane-per-query replay, distinct labels, 64 queries per wave
    ("heap.hip", "heap_replay_lanes_kernelILb1ELb0ELi64ELb0ELb0EE"): (128, 4),
    # one query over a long array: one workgroup of 16 waves, heap in registers
    ("heap.hip", "flat_top_one_kernelILb1EE"): (128, 4),
    ("rescore.hip", "rescore_staged_kernelILi32EE"): (128, 4),
    # table build, the head of the front stream's chain: float queries (GloVe-shaped) and the float64
    # form of every ROTATED index (configs[0], [2], [4]) — round 4 shipped the latter with 272 B/lane
    # of scratch (a dynamically indexed diff[32])
    ("tables.hip", "build_tables_kernelIfLb1EE"): (128, 4),
    ("tables.hip", "build_tables_kernelIdLb1EE"): (128, 4),
    # the lazy lane replay of long rows (FlatTop, configs[2]) and the wave-per-query replays
    ("heap.hip", "heap_replay_lanes_kernelILb1ELb0ELi64ELb1ELb0EE"): (80, 4),
    # labels that repeat (IVF.build(n_probes >= 2), the reference's default): the TWIN form, staged and lazy
    # (one wave per SIMD is all a replay gets: 157 - 314 waves on 1024 SIMDs)
    ("heap.hip", "heap_replay_lanes_kernelILb1ELb0ELi64ELb0ELb1EE"): (144, 3),
    ("heap.hip", "heap_replay_lanes_kernelILb1ELb0ELi64ELb1ELb1EE"): (96, 4),
    # the register heap (one query per wave; round 6): two / four / eight nodes per lane, no LDS
    ("heap.hip", "heap_replay_pair_kernelILb1ELi1EE"): (48, 8),
    ("heap.hip", "heap_replay_pair_kernelILb1ELi2EE"): (64, 7),
    ("heap.hip", "heap_replay_pair_kernelILb1ELi4EE"): (112, 4),
    ("heap.hip", "heap_replay_packed_kernelILb1ELb0EE"): (64, 8),
    ("heap.hip", "heap_replay_packed_kernelILb1ELb1EE"): (64, 8),
}


# every kernel of these files: no scratch, no VGPR spills (rotated / 100M x 128 / build_probes = 2 indexes
# launch instantiations the default batch does not)
NO_SCRATCH_FILES = ("adc_scan.hip", "plain_scan.hip", "tables.hip", "rescore.hip", "heap.hip")
# SGPR spills tolerated (to VGPR lanes, not memory): the duplicate-test lane replay of build_probes >= 2
# ... and the SSE-order form of the plain kernel (the reference's non-AVX module order; no BASELINE config)
# ... and its TWIN form: kernel arguments that are only needed behind the replay loop, parked in VGPR lanes in the
# prologue and fetched back in the epilogue (no v_readlane / v_writelane inside the loop: read off the ISA)
SGPR_SPILLS_OK = {"heap_replay_lanes_kernelILb1ELb1ELi32ELb0ELb0EE": 48, "heap_replay_lanes_kernelILb0ELb1ELi32ELb0ELb0EE": 48,
                  "heap_replay_lanes_kernelILb1ELb0ELi64ELb0ELb1EE": 24, "heap_replay_lanes_kernelILb0ELb0ELi64ELb0ELb1EE": 24,
                  "heap_replay_lanes_kernelILb1ELb0ELi64ELb1ELb1EE": 24, "heap_replay_lanes_kernelILb0ELb0ELi64ELb1ELb1EE": 24,
                  "scan_plain_wave_kernelILi26ELb0EE": 16,
                  # the register heap with four / eight nodes per lane (heaps of 130 ... 513 entries): its lane masks (one
                  # per group and role) outnumber the SGPRs; the two-node form of the common heaps (<= 129) has none
                  "heap_replay_pair_kernelILb1ELi2EE": 16, "heap_replay_pair_kernelILb0ELi2EE": 16,
                  "heap_replay_pair_kernelILb1ELi4EE": 96, "heap_replay_pair_kernelILb0ELi4EE": 96}


def _usage(fname):
    r = subprocess.run([HIPCC] + FLAGS + [fname], cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = {}
    cur = None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


@pytest.fixture(scope="module")
def usage():
    if not shutil.which(HIPCC) and not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    files = sorted({f for f, _ in PINNED} | set(NO_SCRATCH_FILES))
    with ThreadPoolExecutor(max_workers=4) as ex:
        return dict(zip(files, ex.map(_usage, files)))


@pytest.mark.parametrize("key", sorted(PINNED))
def test_no_scratch_and_occupancy(usage, key):
    fname, sub = key
    max_vgpr, min_waves = PINNED[key]
    hits = {k: v for k, v in usage[fname].items() if sub in k}
    assert hits, f"no kernel matching {sub} in {fname}: " + ", ".join(sorted(usage[fname]))[:2000]
    for name, u in hits.items():
        assert u.get("ScratchSize") == 0, (name, u)
        ok_sgpr = max([n for sub_, n in SGPR_SPILLS_OK.items() if sub_ in name] + [0])
        assert u.get("VGPRs Spill", 0) == 0 and u.get("SGPRs Spill", 0) <= ok_sgpr, (name, u)
        assert u["VGPRs"] + u.get("AGPRs", 0) <= max_vgpr, (name, u)
        assert u["Occupancy"] >= min_waves, (name, u)


def test_no_kernel_of_the_hot_files_uses_scratch(usage):
    """Every instantiation of the scan, table, rescoring and replay kernels, not only the default ones."""
    bad = {}
    for f in NO_SCRATCH_FILES:
        for k, v in usage[f].items():
            ok_sgpr = max([n for sub, n in SGPR_SPILLS_OK.items() if sub in k] + [0])
            if v.get("ScratchSize", 0) != 0 or v.get("VGPRs Spill", 0) != 0 or v.get("SGPRs Spill", 0) > ok_sgpr:
                bad[k] = {x: v.get(x) for x in ("ScratchSize", "VGPRs Spill", "SGPRs Spill")}
    assert not bad, bad


def test_twin_replay_parks_its_sgprs_outside_the_loops():
    """The TWIN form of the lane replay reports SGPR spills: kernel arguments that are only needed behind the replay
    loop.  They must stay there — no v_readlane / v_writelane in any basic block of a loop (read off the ISA)."""
    if not shutil.which(HIPCC) and not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "heap.s")
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                            "--cuda-device-only", "-S", "-o", out, "heap.hip"], cwd=CSRC, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        asm = open(out).read()
    seen = 0
    for name in re.findall(r"^(_Z24heap_replay_lanes_kernelILb[01]ELb0ELi64ELb[01]ELb1EE\S*):", asm, flags=re.M):
        body = asm[asm.index("\n" + name + ":"):]
        body = body[:body.index("s_endpgm")]
        in_loop = False
        for line in body.split("\n"):
            m = re.match(r"^\.LBB\d+_\d+:\s*(;.*)?", line)
            if m:
                in_loop = "Loop" in (m.group(1) or "")
            elif in_loop:
                assert "v_readlane" not in line and "v_writelane" not in line, (name[:60], line.strip())
        seen += 1
    assert seen == 4
