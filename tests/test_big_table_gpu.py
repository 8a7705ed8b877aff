"""GPU suite (-m gpu): BASELINE configs 4 and 5 AT THEIR SIZE -- a 1 B-row move table (8 GB as regular-thresholds,
6 GB on disk as blocked-thresholds) -- against the oracle on the same image.

What only this size exercises (reference: include/move_row.hpp:232-243, 36-bit ids; src/move_structure.cpp:59-87 LF_move;
src/move_structure_search.cpp:340-352 count): row ids that need 30 bits, byte offsets into the table beyond 2^32 (32-bit
row INDEXES times 8 bytes), blocked ids with ~954 id blocks, the row-start checkpoints of the count query over 2^30
rows, and a table ~30 x the Infinity Cache.  The oracle walks slices of the batch (first / middle / last 2 k reads):
PMLs, counts, matched lengths and the fast-forward / scan counters must be equal bit for bit.

Host memory: the generator keeps ~22 GB of arrays for 1 B rows next to the 8 GB image and the oracle copies the rows
(8 GB): ~40 GB peak.  Skipped when the machine has less than 96 GB or the GPU less than 24 GB free.
"""
import os

import numpy as np
import pytest

from conftest import vector_kernel

pytestmark = pytest.mark.gpu

ROWS = 1_000_000_000
N_READS, READ_LEN = 1_000_000, 150


def _mem_ok():
    try:
        kb = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1])
        lim = open("/sys/fs/cgroup/memory.max").read().strip() if os.path.exists("/sys/fs/cgroup/memory.max") else "max"
        avail = kb * 1024 if lim == "max" else min(kb * 1024, int(lim))
        return avail >= 96 << 30
    except Exception:
        return True


def _slices(n):
    return [(0, 2000), (n // 2 - 1000, n // 2 + 1000), (n - 2000, n)]


@pytest.mark.parametrize("mode", [6, 8])
def test_one_billion_row_table_vs_oracle(built_lib, mode):
    import torch
    import movi_amd
    from oracle.oracle import Oracle
    from tools import synth
    if not _mem_ok():
        pytest.skip("needs ~40 GB of host memory")
    free, _ = torch.cuda.mem_get_info()
    if free < 40 << 30:
        pytest.skip("needs 40 GB of free HBM")
    six = synth.synth_index(ROWS, mode=mode, seed=20260529)
    img = six.image()
    bases, offs = synth.synth_reads(six, N_READS, READ_LEN, seed=20260530, sub_rate=0.01, n_rate=0.001)
    del six                                              # the generator's 22 GB of arrays
    gpu = movi_amd.MoveIndex.from_image(img)
    assert gpu.desc.r == ROWS and gpu.desc.mode == mode
    if mode == 8:
        assert gpu.desc.n_blocks >= 900                  # ~954 id blocks of 2^20 rows (or more, halved adaptively)

    # ---- PML: the whole batch in one launch (the config-4 shape; "host_autopin" off: with it a call this big would be cut
    # into overlapped pieces), then the slices on their own for the counters
    gpu.set_option("host_autopin", 0)
    out, st = gpu.query_pml_packed(bases, offs)
    assert st.errors == 0 and st.bases == bases.size
    li = gpu.last_launch()
    # the first query built the look-ahead copy (16 GB) and walks on it with pair-shared gathers (a table beyond the TLBs' reach)
    # (round 6: a host call of this size brings reset masks down -- RING = 2; vector_kernel reads that as the layout's vector kernel)
    assert vector_kernel(li["kernel"]) == "pml_kernel_flatp<6, unsigned int, 0, 0, 0, 1, 1, 1, 0>" and li["idx64"] == 0, li
    assert li["ahead"] == 1 and li["waves_per_cu"] == 9 and gpu.info("ahead_rows_bytes") >= 16e9
    gpu.set_option("host_autopin", 1)
    out2, st2 = gpu.query_pml_packed(bases, offs)          # ... and the same call cut into overlapped pieces: identical
    assert (out2 == out).all() and (st2.fast_forwards, st2.scans) == (st.fast_forwards, st.scans)
    del out2
    # ---- count: the config-5 query
    m, c, cst = gpu.query_count_packed(bases, offs)
    assert cst.errors == 0
    cpu = Oracle(img)
    del img
    for lo, hi in _slices(N_READS):
        sb = bases[int(offs[lo]): int(offs[hi])]
        so = offs[lo: hi + 1] - offs[lo]
        exp, eff, esc = cpu.pml_batch(sb, so, threads=8)
        assert (out[int(offs[lo]): int(offs[hi])] == exp).all(), (mode, lo)
        sout, sst = gpu.query_pml_packed(sb, so)
        assert (sout == exp).all()
        assert (sst.fast_forwards, sst.scans, sst.errors) == (eff, esc, 0), (mode, lo)
        em, ec = cpu.count_batch(sb, so, threads=8)
        assert (m[lo:hi] == em).all() and (c[lo:hi] == ec).all(), (mode, lo)
    # walks really spread over the whole table: matches on a 1 B-row table end in rows far beyond 2^29
    assert int(c.max()) >= 1 and int(m.max()) > 20
    # ---- the 64-bit row-index instantiations on the same table (what a table of 2^32 rows and more would run)
    gpu.set_option("idx64", 1)
    lo, hi = _slices(N_READS)[1]
    sb = bases[int(offs[lo]): int(offs[hi])]
    so = offs[lo: hi + 1] - offs[lo]
    exp, eff, esc = cpu.pml_batch(sb, so, threads=8)
    sout, sst = gpu.query_pml_packed(sb, so)
    assert gpu.last_launch()["idx64"] == 1
    assert (sout == exp).all() and (sst.fast_forwards, sst.scans) == (eff, esc)
    gpu.close()
    cpu.close()
