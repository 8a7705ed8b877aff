"""GPU suite (-m gpu): the bench's own workloads at their size, held to the oracle.

* c2: the 14 M-row real-BWT pangenome index of `bench.py` (built once per box by tools/build_index and cached -- or found in a
  .bench_cache/ that travelled with the tree), 1 M x 150 bp reads through movi_pml_device: 20 000 reads of the batch against
  the oracle, PMLs and counters, on every table layout the walk can run on.
* c3 as BASELINE.json words it -- "100 k x 10 kbp ... PML + --classify": the fused classification kernels at that size
  (movi_pml_classify_device with and without the PML vector: CLS = 1 / 2), bins against Classifier::classify restated over the
  oracle's PMLs on slices (src/classifier.cpp:99-143, src/read_processor.cpp:565-578), the bins-only launch against the
  vector launch on all 100 000 reads.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, classify_py

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pangenome(built_lib):
    """(index directory, c2 reads file): bench.py's c2 workload, built on first use (~2 min on a fresh box)."""
    sys.path.insert(0, ROOT)
    import bench
    wl = dict(bench.WORKLOADS["c2"])
    idx_dir, reads_file = bench.ensure_pangenome(wl, 1, 0, lambda: None)
    return idx_dir, reads_file


def _load(idx_dir):
    import movi_amd
    from oracle.oracle import Oracle
    img = np.fromfile(os.path.join(idx_dir, "index.movi"), np.uint8)
    return movi_amd.MoveIndex.from_image(img), Oracle(img)


@pytest.mark.timeout(1800)
def test_c2_pangenome_batch_vs_oracle(pangenome):
    import torch
    idx_dir, reads_file = pangenome
    gpu, cpu = _load(idx_dir)
    n, L = 1_000_000, 150
    bases = np.fromfile(reads_file, np.uint8, count=n * L)
    offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    dev = torch.device("cuda", 0)
    d_bases = torch.from_numpy(bases).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
    d_out = torch.empty(n * L, dtype=torch.int16, device=dev)
    d_err = torch.zeros(n, dtype=torch.uint8, device=dev)
    lo, cnt = 490_000, 20_000                                  # 20 k reads from the middle of the batch
    sb, so = bases[lo * L: (lo + cnt) * L], offs[: cnt + 1]
    exp, eff, esc = cpu.pml_batch(sb, so, threads=8)
    for ahead, variant in ((1, -1), (0, -1), (1, 14)):
        gpu.set_option("ahead_rows", ahead)
        gpu.set_option("pml_variant", variant)
        gpu.pml_device(d_bases.data_ptr(), d_offs.data_ptr(), n, n * L, d_out.data_ptr(), d_err.data_ptr())
        torch.cuda.synchronize()
        li = gpu.last_launch()
        assert li["ahead"] == ahead and li["variant"] == 14, li
        got = d_out[lo * L: (lo + cnt) * L].cpu().numpy().view(np.uint16)
        assert (got == exp).all(), (ahead, variant)
        assert int(d_err.sum().item()) == 0
        # the slice alone: its counters equal the oracle's
        d_so = torch.from_numpy(so.view(np.int64).copy()).to(dev)
        gpu.pml_device(d_bases.data_ptr() + lo * L, d_so.data_ptr(), cnt, cnt * L, d_out.data_ptr(), d_err.data_ptr())
        st = gpu.last_stats()
        assert (st.fast_forwards, st.scans, st.errors) == (eff, esc, 0), (ahead, variant)
        assert (d_out[: cnt * L].cpu().numpy().view(np.uint16) == exp).all()
    gpu.set_option("pml_variant", -1)
    gpu.close()
    cpu.close()


@pytest.mark.timeout(2400)
def test_c3_classify_at_size(pangenome):
    """BASELINE config 3: 100 k x 10 kbp, PML + --classify."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    idx_dir, _ = pangenome
    gpu, cpu = _load(idx_dir)
    n, L = 100_000, 10_000
    tool = os.path.join(ROOT, "tools", "build_index")
    text = os.path.join(idx_dir, "text.bin")
    PG = bench.PG_C2
    if not os.path.exists(text):
        subprocess.check_call([tool, "pangenome", str(PG["anc"]), str(PG["genomes"]), str(PG["snp"]), str(PG["seed"]), "6", idx_dir,
                               "text-only"], stderr=subprocess.DEVNULL)
    rf = os.path.join(idx_dir, "reads_%dx%d_%g.bin" % (n, L, 0.08))
    if not os.path.exists(rf):
        subprocess.check_call([tool, "reads", text, str(n), str(L), "0.08", str(PG["seed"]), rf + ".tmp"])
        os.rename(rf + ".tmp", rf)
    bases = np.fromfile(rf, np.uint8, count=n * L)
    dev = torch.device("cuda", 0)
    d_bases = torch.from_numpy(bases).to(dev)
    d_offs = torch.from_numpy((np.arange(n + 1, dtype=np.uint64) * np.uint64(L)).view(np.int64)).to(dev)
    d_out = torch.zeros(n * L, dtype=torch.int16, device=dev)
    d_err = torch.zeros(n, dtype=torch.uint8, device=dev)
    bin_width, thr = 150, 8
    res = {}
    # vector + bins fused into the walk (CLS = 1), the walk followed by the streaming pass over the vectors (what the policy
    # picks for long reads: classify_wave_kernel), and bins without the vector (CLS = 2)
    for name, with_vector, fused, cls in (("fused", True, 1, 1), ("two_pass", True, -1, 0), ("bins_only", False, -1, 2)):
        gpu.set_option("classify_fused", fused)
        d_a = torch.full((n,), -1, dtype=torch.int32, device=dev)
        d_b = torch.full((n,), -1, dtype=torch.int32, device=dev)
        d_s = torch.full((n,), -1, dtype=torch.int64, device=dev)
        d_out.zero_()
        gpu.pml_classify_device(d_bases.data_ptr(), d_offs.data_ptr(), n, n * L, bin_width, thr,
                                d_out.data_ptr() if with_vector else 0, d_a.data_ptr(), d_b.data_ptr(), d_s.data_ptr(), d_err.data_ptr())
        torch.cuda.synchronize()
        assert gpu.last_launch()["kernel"].startswith("pml_kernel_flatp<6, unsigned int, %d," % cls), name
        assert int(d_err.sum().item()) == 0
        res[name] = (d_a.cpu().numpy(), d_b.cpu().numpy(), d_s.cpu().numpy())
        if name == "fused":
            got_fused = d_out.cpu().numpy().view(np.uint16).copy()
        elif name == "two_pass":
            assert (got_fused == d_out.cpu().numpy().view(np.uint16)).all()
    # all 100 000 reads: the three launches agree
    for other in ("two_pass", "bins_only"):
        for x, y in zip(res["fused"], res[other]):
            assert (x == y).all(), other
    a, b, sm = res["fused"]
    assert ((a + b) == L // bin_width).all()                   # 66 bins of 150, the last one absorbs the remainder (classifier.cpp:110-115)
    # bins of slices against Classifier::classify over the ORACLE's PMLs, and the vectors themselves
    got = got_fused
    for lo in (0, 49_950, n - 100):
        sb = bases[lo * L: (lo + 100) * L]
        so = np.arange(101, dtype=np.uint64) * np.uint64(L)
        exp, _, _ = cpu.pml_batch(sb, so, threads=8)
        assert (got[lo * L: (lo + 100) * L] == exp).all(), lo
        for i in range(100):
            found, avg, ea, eb = classify_py(exp[i * L: (i + 1) * L], thr, bin_width)
            assert (int(a[lo + i]), int(b[lo + i])) == (ea, eb) and int(sm[lo + i]) == round(avg * (ea + eb)), (lo, i)
    gpu.close()
    cpu.close()


@pytest.mark.timeout(1800)
def test_real_bwt_beyond_every_cache_vs_oracle(built_lib):
    """A REAL BWT that fits no cache, under the driver's eye (round 6): bench.py's `c2mid` -- 64 genomes of a 2.5 Mbp ancestor with 1.3 %
    SNPs, 39.5 M rows = 316 MB, its look-ahead copy 632 MB and its deep rows 842 MB, all beyond the 256 MB Infinity Cache (built by
    tools/build_index in a minute or two: an index of this size does not fit the 512 MB a tree may carry to the GPU box).  The DEFAULT policy's launches of the whole 1 M x 150 bp batch -- device entry
    point: deep rows, vector through fused reset masks; the same with the walk's own packer; on the look-ahead rows; the count query --
    with 20 000-read slices from the start, the middle and the end held to the oracle: PMLs, fast-forward / scan counters, matched
    lengths and counts; the kernels the policy picked asserted by name.
    Reference: src/move_structure.cpp:59-87 (LF_move), src/move_structure_query.cpp:513-601, src/move_structure_search.cpp:340-352."""
    import torch
    import movi_amd
    sys.path.insert(0, ROOT)
    import bench
    from oracle.oracle import Oracle
    wl = dict(bench.WORKLOADS["c2mid"])
    idx_dir, reads_file = bench.ensure_pangenome(wl, 1, 0, lambda: None)
    img = np.fromfile(os.path.join(idx_dir, "index.movi"), np.uint8)
    n, L = wl["reads"], wl["read_len"]
    bases = np.fromfile(reads_file, np.uint8, count=n * L)
    offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    rows = int(gpu.desc.r)
    assert 30_000_000 < rows < 48_000_000, rows
    gpu.prepare(gpu.PREPARE_PML | gpu.PREPARE_COUNT)
    assert gpu.info("deep_rows_bytes") > 700e6 and gpu.info("ahead_rows_bytes") > 500e6 and gpu.info("ahead_no_ff") >= 0.67
    dev = torch.device("cuda", 0)
    d_bases = torch.from_numpy(bases).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
    d_out = torch.empty(n * L, dtype=torch.int16, device=dev)
    d_err = torch.zeros(n, dtype=torch.uint8, device=dev)
    slices = [(0, 20_000), (n // 2 - 10_000, n // 2 + 10_000), (n - 20_000, n)]
    expected = []
    for lo, hi in slices:
        expected.append(cpu.pml_batch(bases[lo * L: hi * L], offs[: hi - lo + 1], threads=16))
    for name, opts, kernel, ahead in (("default", {}, "pml_kernel_flatp<6, unsigned int, 0, 0, 0, 1, 2, 0, 2>", 2),
                                      ("packer", {"pml_via_mask": 0}, "pml_kernel_flatp<6, unsigned int, 0, 0, 0, 1, 2, 0, 0>", 2),
                                      ("look-ahead rows", {"deep": 0}, "pml_kernel_flatp<6, unsigned int, 0, 0, 0, 1, 1, 0, 2>", 1)):
        for k, v in opts.items():
            gpu.set_option(k, v)
        d_out.fill_(-1)
        gpu.pml_device(d_bases.data_ptr(), d_offs.data_ptr(), n, n * L, d_out.data_ptr(), d_err.data_ptr())
        torch.cuda.synchronize()
        li = gpu.last_launch()
        assert li["kernel"] == kernel and li["ahead"] == ahead and li["waves_per_cu"] == (13 if ahead == 2 else 9), (name, li)
        assert int(d_err.sum().item()) == 0
        for (lo, hi), (exp, eff, esc) in zip(slices, expected):
            got = d_out[lo * L: hi * L].cpu().numpy().view(np.uint16)
            assert (got == exp).all(), (name, lo)
            d_so = torch.from_numpy(offs[: hi - lo + 1].view(np.int64).copy()).to(dev)
            gpu.pml_device(d_bases.data_ptr() + lo * L, d_so.data_ptr(), hi - lo, (hi - lo) * L, d_out.data_ptr(), d_err.data_ptr())
            st = gpu.last_stats()
            assert (st.fast_forwards, st.scans, st.errors) == (eff, esc, 0), (name, lo)
        for k in opts:
            gpu.set_option(k, -1)
    # the count query (the state machine on the plain rows: 302 MB)
    m, c, cst = gpu.query_count_packed(bases, offs)
    assert cst.errors == 0 and gpu.last_launch()["kernel"] == "zml_kernel_flat<6, unsigned int, 0, 0, 0, 1>"
    for lo, hi in slices:
        em, ec = cpu.count_batch(bases[lo * L: hi * L], offs[: hi - lo + 1], threads=16)
        assert (m[lo:hi] == em).all() and (c[lo:hi] == ec).all(), lo
    gpu.close()
    cpu.close()
