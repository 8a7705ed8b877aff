"""GPU suite, slow part (-m gpu with MOVI_SLOW_TESTS=1; tools/r05_real_bwt.sh): the real BWTs behind README's / DESIGN's
"c4real", "c4real2" and "c4big" numbers -- 113 M, 226 M and 676 M rows, beyond the Infinity Cache (and, for the second one's look-ahead copy,
beyond the TLBs' reach) -- against the oracle AT THEIR SIZE.  bench.py's own workloads: the index is built on first use by
tools/build_index (3 - 10 min of host time, 16 - 35 GB of host memory; cached under $MOVI_BENCH_CACHE), which is why these
tests are not part of the default suite: the driver's GPU step has 20 minutes for everything.

What they hold to the oracle: the DEFAULT policy's launch of the whole 1.25 M-read batch (look-ahead rows; pair-shared gathers on
the 226 M-row table's 3.6 GB copy) -- PML vectors of 20 k-read slices from the start, the middle and the end of the batch, the
slices' fast-forward / scan counters, and the count query (matched lengths and counts) on the same slices.
Reference: src/move_structure.cpp:59-87 (LF_move), src/move_structure_query.cpp:513-601, src/move_structure_search.cpp:340-352.
"""
import os

import numpy as np
import pytest

from conftest import vector_kernel

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


def _slices(n, k=20_000):
    return [(0, k), (n // 2 - k // 2, n // 2 + k // 2), (n - k, n)]


@pytest.mark.parametrize("workload", ["c4real", "c4real2", "c4big"])
def test_real_bwt_at_size_vs_oracle(built_lib, workload):
    import movi_amd
    import bench
    from oracle.oracle import Oracle
    wl = dict(bench.WORKLOADS[workload])
    idx_dir, reads_file = bench.ensure_pangenome(wl, 1, 0, lambda: None)
    img = np.fromfile(os.path.join(idx_dir, "index.movi"), np.uint8)
    n, L = wl["reads"], wl["read_len"]
    bases = np.fromfile(reads_file, np.uint8, count=n * L)
    offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    gpu = movi_amd.MoveIndex.from_image(img)
    rows = int(gpu.desc.r)
    assert rows > {"c4real": 100_000_000, "c4real2": 200_000_000, "c4big": 400_000_000}[workload]
    gpu.prepare(gpu.PREPARE_PML | gpu.PREPARE_COUNT)
    gpu.set_option("host_autopin", 0)                    # the whole batch in one launch (the bench's shape)
    out, st = gpu.query_pml_packed(bases, offs)
    li = gpu.last_launch()
    assert st.errors == 0 and st.bases == bases.size
    pair = 1 if rows * 16 >= 2 << 30 else 0             # pair-shared gathers: walked tables of 2 GB and more
    # (round 6: a host call of this size brings reset masks down -- RING = 2; vector_kernel reads that as the layout's vector kernel)
    assert vector_kernel(li["kernel"]) == "pml_kernel_flatp<6, unsigned int, 0, 0, 0, 1, 1, %d, 0>" % pair and li["ahead"] == 1 and li["waves_per_cu"] == 9, li
    m, c, cst = gpu.query_count_packed(bases, offs)
    cli = gpu.last_launch()
    cpair = 1 if rows * 8 >= 2 << 30 else 0             # the count query's state machine: pairs on plain rows of 2 GB and more
    assert cst.errors == 0 and cli["kernel"] == "zml_kernel_flat<6, unsigned int, 0, 0, %d, 1>" % cpair, cli
    no_ff = gpu.info("ahead_no_ff")
    assert no_ff >= (0.67 if workload != "c4big" else 0.4), no_ff   # (0.83 on the c2 pangenome, 0.75 at 1 % SNPs; 0.51 on random run sequences)
    cpu = Oracle(img)
    for lo, hi in _slices(n):
        sb, so = bases[lo * L: hi * L], offs[: hi - lo + 1]
        exp, eff, esc = cpu.pml_batch(sb, so, threads=16)
        assert (out[lo * L: hi * L] == exp).all(), (workload, lo)
        sout, sst = gpu.query_pml_packed(sb, so)
        assert (sout == exp).all() and (sst.fast_forwards, sst.scans, sst.errors) == (eff, esc, 0), (workload, lo)
        em, ec = cpu.count_batch(sb, so, threads=16)
        assert (m[lo:hi] == em).all() and (c[lo:hi] == ec).all(), (workload, lo)
    # count_kernel_v0 on the look-ahead rows (the A/B kernel) gives the same
    gpu.set_option("count_variant", 0)
    lo, hi = _slices(n)[1]
    sm, sc, _ = gpu.query_count_packed(bases[lo * L: hi * L], offs[: hi - lo + 1])
    assert gpu.last_launch()["kernel"].startswith("count_kernel_v0<6, ") and (sm == m[lo:hi]).all() and (sc == c[lo:hi]).all()
    gpu.close()
    cpu.close()
