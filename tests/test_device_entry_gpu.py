"""GPU suite (-m gpu): the contracts of the *_device entry points and of the multi-GPU index replication.

* the segment plan sizes its scratch from the caller's n_bases: a batch whose offsets break `offsets[0] == 0,
  offsets[n_reads] <= n_bases` must still be answered correctly (the plan stands down on the device);
* "seg_probe" = 2: the caller states the verdict, nothing is read back, same answers either way;
* the overlapped classify host path with chunks that qualify for segmentation (each chunk in flight brings its own
  segment workspace);
* movi_last_launch names the kernel the policy picked;
* movi_index_replicate / movi_index_load_replicated (RCCL broadcast, one rank on a 1-GPU box) give handles that answer
  like movi_index_load's.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_gpu_parity import _pinned_copy, mutated_reads, pack

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine6(built_lib, golden_image):
    import movi_amd
    from oracle.oracle import Oracle
    img = golden_image(6)
    return movi_amd.MoveIndex.from_image(img), Oracle(img)


def _ref():
    from oracle import build_index as B
    return B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]


def test_device_entry_with_offsets_that_do_not_start_at_zero(engine6):
    """A sub-batch passed as a window into a larger offsets array (offsets[0] != 0) with n_bases = its own span: the
    segment plan would index its checkpoints past their end; it must stand down (device-side check) and the reads be
    walked by one lane each -- PML and ZML, same answers as the oracle."""
    import torch
    gpu, cpu = engine6
    rng = np.random.default_rng(9100)
    reads = mutated_reads(rng, _ref(), 40, 3000, 6000)
    bases, offs = pack(reads)
    skip = 7                                              # the window: reads [skip, n)
    n = len(reads) - skip
    dev = torch.device("cuda", 0)
    d_bases = torch.from_numpy(bases.copy()).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64).copy()).to(dev)
    span = int(offs[-1] - offs[skip])
    exp, _, _ = cpu.pml_batch(bases, offs, threads=4)
    expz = cpu.zml_batch(bases, offs, threads=4)
    gpu.set_option("seg_len", 64)
    try:
        for probe in (0, 1):
            gpu.set_option("seg_probe", probe)
            for zml in (False, True):
                d_out = torch.zeros(bases.size, dtype=torch.int16, device=dev)   # indexed by the ABSOLUTE offsets of the window
                fn = gpu.zml_device if zml else gpu.pml_device
                fn(d_bases.data_ptr(), d_offs.data_ptr() + 8 * skip, n, span, d_out.data_ptr())
                torch.cuda.synchronize()
                st = gpu.last_stats()
                assert st.errors == 0 and st.rewalked == 0
                got = d_out.cpu().numpy().view(np.uint16)
                want = expz if zml else exp
                assert (got[int(offs[skip]):] == want[int(offs[skip]):]).all(), (probe, zml)
                assert not got[: int(offs[skip])].any()                          # nothing written outside the window
        # the same window passed properly (rebased pointers and offsets): cut into segments, same answers
        gpu.set_option("seg_probe", 0)
        rel = (offs[skip:] - offs[skip]).astype(np.uint64)
        d_rel = torch.from_numpy(rel.view(np.int64).copy()).to(dev)
        d_out = torch.zeros(span, dtype=torch.int16, device=dev)
        gpu.pml_device(d_bases.data_ptr() + int(offs[skip]), d_rel.data_ptr(), n, span, d_out.data_ptr())
        torch.cuda.synchronize()
        assert gpu.last_stats().segments > n and gpu.last_launch()["segmented"] == 1
        assert (d_out.cpu().numpy().view(np.uint16) == exp[int(offs[skip]):]).all()
        # n_bases smaller than the batch really is: stands down as well
        d_out.zero_()
        gpu.pml_device(d_bases.data_ptr() + int(offs[skip]), d_rel.data_ptr(), n, span // 2, d_out.data_ptr())
        torch.cuda.synchronize()
        assert gpu.last_stats().errors == 0
        assert (d_out.cpu().numpy().view(np.uint16) == exp[int(offs[skip]):]).all()
    finally:
        gpu.set_option("seg_len", 2048)
        gpu.set_option("seg_probe", 1)


def test_caller_supplied_segment_verdict_keeps_the_call_asynchronous(engine6):
    """seg_probe = 2: no probe, no read-back; "seg_verdict" 1 cuts the batch, 0 does not.  Identical answers; and the
    launch really is asynchronous: it can be captured into a HIP graph (a stream synchronise inside would fail the
    capture)."""
    import torch
    gpu, cpu = engine6
    rng = np.random.default_rng(9200)
    reads = mutated_reads(rng, _ref(), 64, 5000, 7000)
    bases, offs = pack(reads)
    n = len(reads)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=4)
    expz = cpu.zml_batch(bases, offs, threads=4)
    dev = torch.device("cuda", 0)
    d_bases = torch.from_numpy(bases.copy()).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64).copy()).to(dev)
    gpu.set_option("seg_probe", 2)
    try:
        for verdict in (0, 1):
            gpu.set_option("seg_verdict", verdict)
            for zml in (False, True):
                d_out = torch.zeros(bases.size, dtype=torch.int16, device=dev)
                (gpu.zml_device if zml else gpu.pml_device)(d_bases.data_ptr(), d_offs.data_ptr(), n, bases.size, d_out.data_ptr())
                torch.cuda.synchronize()
                st = gpu.last_stats()
                assert (st.segments > n) == bool(verdict), (verdict, zml, st.segments)
                assert gpu.last_launch()["segmented"] == verdict
                assert (d_out.cpu().numpy().view(np.uint16) == (expz if zml else exp)).all()
                if not zml:
                    assert (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0)
        # graph capture of the segmented launch (workspace already sized by the calls above)
        gpu.set_option("seg_verdict", 1)
        s = torch.cuda.Stream()
        d_out = torch.zeros(bases.size, dtype=torch.int16, device=dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            gpu.pml_device(d_bases.data_ptr(), d_offs.data_ptr(), n, bases.size, d_out.data_ptr(), 0, s.cuda_stream)   # warm
            s.synchronize()
            d_out.zero_()
            s.synchronize()
            g.capture_begin(capture_error_mode="relaxed")
            gpu.pml_device(d_bases.data_ptr(), d_offs.data_ptr(), n, bases.size, d_out.data_ptr(), 0, s.cuda_stream)
            g.capture_end()
        assert int(d_out.abs().sum().item()) == 0           # captured, not run
        g.replay()
        torch.cuda.synchronize()
        assert (d_out.cpu().numpy().view(np.uint16) == exp).all()
    finally:
        gpu.set_option("seg_probe", 1)
        gpu.set_option("seg_verdict", 0)


def test_overlapped_classify_with_segmented_chunks(engine6):
    """movi_pml_classify_host, page-locked bases, chunks small enough that several are in flight and long reads so that
    every chunk is walked segment-parallel: each chunk must use its own slot's segment workspace (they used to share the
    handle's).  Bins equal the one-lane-per-read bins, repeatedly."""
    gpu, cpu = engine6
    rng = np.random.default_rng(9300)
    reads = mutated_reads(rng, _ref(), 260, 300, 1500)     # mean <= 2048: the host path overlaps; >= 2 x seg_len: every chunk is cut
    bases, offs = pack(reads)
    gpu.set_option("seg_len", 0)
    exp = gpu.classify_packed(bases, offs, 150, 8)
    pb = _pinned_copy(bases)
    gpu.set_option("seg_len", 64)
    gpu.set_option("seg_probe", 0)
    gpu.set_option("pipe_chunk_bases", 20_000)             # ~22 reads per chunk, a dozen chunks, 3 in flight
    try:
        for rep in range(4):
            got = gpu.classify_packed(pb, offs, 150, 8)
            assert all((x == y).all() for x, y in zip(got, exp)), rep
    finally:
        gpu.set_option("seg_len", 2048)
        gpu.set_option("seg_probe", 1)
        gpu.set_option("pipe_chunk_bases", 0)


def test_last_launch_reports_the_policy(engine6):
    gpu, _ = engine6
    bases, offs = pack([b"ACGTACGTAC" * 20] * 300)
    gpu.query_pml_packed(bases, offs)
    li = gpu.last_launch()
    # the default walk of a batch of short reads (round 6): reads staged through LDS (a small batch: uncapped, 336 bases per lane), on the
    # DEEP rows (a small table of real text), the vector through reset masks that every wavefront expands itself (RING = 2)
    assert li["kernel"] == "pml_kernel_flatp<6, unsigned int, 0, 0, 0, 1, 2, 0, 2>" and li["variant"] == 14
    assert li["block_threads"] == 64 and li["waves_per_cu"] == 0 and li["segmented"] == 0 and li["staged"] == 336 and li["ahead"] == 2
    gpu.set_option("pml_via_mask", 0)
    gpu.set_option("host_masks", 0)
    gpu.query_pml_packed(bases, offs)
    assert gpu.last_launch()["kernel"] == "pml_kernel_flatp<6, unsigned int, 0, 0, 0, 1, 2, 0, 0>"        # ... by the register packer
    gpu.set_option("pml_via_mask", -1)
    gpu.set_option("host_masks", -1)
    gpu.set_option("stage_reads", 0)
    gpu.query_pml_packed(bases, offs)
    assert gpu.last_launch()["kernel"] == "pml_kernel_flatp<6, unsigned int, 0, 0, 0, 0, 0, 0, 0>" and gpu.last_launch()["staged"] == 0
    gpu.set_option("stage_reads", 1)
    gpu.set_option("pml_variant", 1)
    gpu.query_pml_packed(bases, offs)
    assert gpu.last_launch()["kernel"] == "pml_kernel<6, 1, 0>"
    gpu.set_option("pml_variant", -1)
    gpu.query_count_packed(bases, offs)
    assert gpu.last_launch()["kernel"] == "zml_kernel_flat<6, unsigned int, 0, 0, 0, 1>" and gpu.last_launch()["ahead"] == 0   # the lane state machine, on the plain rows (round 5); (a small table)
    gpu.query_zml_packed(bases, offs)
    assert gpu.last_launch()["kernel"] == "zml_kernel_flat<6, unsigned int, 0, 0, 0, 0>"


@pytest.mark.parametrize("mode", [6, 8, 7])
def test_replicated_index_through_rccl(built_lib, golden_image, tmp_path, mode):
    """movi_index_replicate / movi_index_load_replicated on the GPUs this box has (one: a communicator of one rank -- the
    same code path: librccl bound at first use, ncclCommInitAll, grouped ncclBroadcast, per-GPU expansion).  The
    handles answer like movi_index_load's; duplicate or missing devices are refused."""
    import torch
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    img = golden_image(mode) if mode != 7 else B.build_index_from_seqs([_ref()], 7)
    cpu = Oracle(img)
    n_dev = torch.cuda.device_count()
    devices = list(range(n_dev))
    reads = mutated_reads(np.random.default_rng(9400 + mode), _ref(), 200, 20, 400)
    bases, offs = pack(reads)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=4)
    d = tmp_path / "idx"
    d.mkdir()
    (d / "index.movi").write_bytes(bytes(img))
    for handles in (movi_amd.MoveIndex.replicate_image(img, devices), movi_amd.MoveIndex.load_replicated(str(d), devices)):
        assert len(handles) == n_dev
        for h in handles:
            out, st = h.query_pml_packed(bases, offs)
            assert (out == exp).all() and (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0)
            assert h.query_count(reads[:20]) == [cpu.count(r) for r in reads[:20]]
            h.close()
    with pytest.raises(movi_amd.MoviError) as e:
        movi_amd.MoveIndex.replicate_image(img, [0, 0])
    assert e.value.code == -1
    with pytest.raises(movi_amd.MoviError) as e:
        movi_amd.MoveIndex.replicate_image(img, [n_dev])
    assert e.value.code == -5


@pytest.mark.parametrize("mode", [6, 8])
def test_logs_per_base_fastforwards_and_scans(built_lib, golden_image, mode):
    """movi_pml_logs_host (`movi query --logs`): per base, the fast-forwards of the LF to the next base and the rows the
    base's reposition scanned, as MoveQuery::add_fastforward / add_scan collect them in the reference's strand scheduler
    (src/read_processor.cpp:99-121, 586-596) -- against the oracle's restatement, read by read; their sums are the batch
    counters of the ordinary query."""
    import movi_amd
    from oracle.oracle import Oracle
    img = golden_image(mode)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    reads = mutated_reads(np.random.default_rng(9500 + mode), _ref(), 300, 1, 500) + [b"", b"A", b"N", b"AC", b"ACGTN" * 30]
    bases, offs = pack(reads)
    out, ff, sc, st = gpu.query_pml_logs_packed(bases, offs)
    assert gpu.last_launch()["kernel"] == "pml_kernel<6, 0, 0>"
    for i, r in enumerate(reads):
        eo, ef, es = cpu.pml_logs(r)
        a, b = int(offs[i]), int(offs[i + 1])
        assert (out[a:b] == eo).all() and (ff[a:b] == ef).all() and (sc[a:b] == es).all(), i
    plain, pst = gpu.query_pml_packed(bases, offs)
    assert (plain == out).all() and st.scans == pst.scans == int(sc.astype(np.uint64).sum())
    gpu.close()


def test_big_pageable_host_call_page_locks_its_buffers(built_lib):
    """movi_pml_host / movi_zml_host on PAGEABLE buffers of >= 2^27 bases: the call page-locks the caller's buffers for its
    duration and takes the overlapped path ("host_autopin", on by default).  Same PMLs, error bytes and counters as the
    synchronous path; the buffers are ordinary pageable memory again afterwards (a second call, and one with the option
    off, work the same)."""
    import movi_amd
    from tools import synth
    six = synth.synth_index(2_000_000, mode=6, seed=15)
    gpu = movi_amd.MoveIndex.from_image(six.image())
    bases, offs = synth.synth_reads(six, 1_000_000, 150, seed=16, sub_rate=0.01, n_rate=0.001)
    assert bases.size >= 1 << 27
    gpu.set_option("host_autopin", 0)
    exp, est = gpu.query_pml_packed(bases, offs)
    expz, _ = gpu.query_zml_packed(bases, offs)
    em, ec, ecst = gpu.query_count_packed(bases, offs)
    ebins = gpu.classify_packed(bases, offs, 150, 6)
    gpu.set_option("host_autopin", 1)
    m, c, cst = gpu.query_count_packed(bases, offs)
    assert (m == em).all() and (c == ec).all() and (cst.fast_forwards, cst.scans) == (ecst.fast_forwards, ecst.scans)
    assert all((x == y).all() for x, y in zip(gpu.classify_packed(bases, offs, 150, 6), ebins))
    for rep in range(2):
        out = np.full(bases.size, 0xABCD, np.uint16)
        got, st, err, rc = gpu.query_pml_packed(bases, offs, want_err=True, out=out)
        assert rc == 0 and (out == exp).all() and not err.any()
        assert (st.bases, st.fast_forwards, st.scans, st.repositions) == (est.bases, est.fast_forwards, est.scans, est.repositions)
    zout, _ = gpu.query_zml_packed(bases, offs)
    assert (zout == expz).all()
    out[:] = 0                                              # still writable, still ours
    gpu.close()


def test_prepared_handle_queries_build_nothing_and_can_be_captured(built_lib, golden_image):
    """movi_index_prepare (round 5): the derived tables -- top-of-walk table, look-ahead rows, interval table, row-start
    checkpoints -- are built by the call, not inside the first query: afterwards the first movi_pml_device / movi_count_device
    on the handle leave the device's free memory where it was and can be captured into a HIP graph with no warm-up call before
    them (a hipMalloc or a stream synchronise inside would fail the capture); the replayed graphs give the oracle's answers."""
    import torch
    import movi_amd
    from oracle.oracle import Oracle
    img = golden_image(6)
    cpu = Oracle(img)
    rng = np.random.default_rng(9700)
    reads = mutated_reads(rng, _ref(), 900, 1, 300)
    bases, offs = pack(reads)
    n = len(reads)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=4)
    em, ec = cpu.count_batch(bases, offs, threads=4)
    dev = torch.device("cuda", 0)
    d_bases = torch.from_numpy(bases.copy()).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64).copy()).to(dev)
    d_out = torch.zeros(bases.size, dtype=torch.int16, device=dev)
    d_m = torch.zeros(n, dtype=torch.int64, device=dev)
    d_c = torch.zeros(n, dtype=torch.int64, device=dev)
    gpu = movi_amd.MoveIndex.from_image(img)
    assert gpu.info("derived_bytes") == 0
    with pytest.raises(movi_amd.MoviError):
        gpu.prepare(8)                                         # not a MOVI_PREPARE_* bit
    got = gpu.prepare(gpu.PREPARE_PML | gpu.PREPARE_COUNT | gpu.PREPARE_ZML)
    assert got == gpu.info("derived_bytes") and got > 2 * (256 << 20)
    assert gpu.info("kmer_bytes") == 256 << 20 and gpu.info("ftab_bytes") == 256 << 20
    assert gpu.info("ahead_rows_bytes") > 0 and gpu.info("ckpt_bytes") > 0
    assert gpu.prepare() == got                                # idempotent
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    s = torch.cuda.Stream()
    g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        g1.capture_begin(capture_error_mode="relaxed")
        gpu.pml_device(d_bases.data_ptr(), d_offs.data_ptr(), n, bases.size, d_out.data_ptr(), 0, s.cuda_stream)
        g1.capture_end()
        g2.capture_begin(capture_error_mode="relaxed")
        gpu.count_device(d_bases.data_ptr(), d_offs.data_ptr(), n, bases.size, d_m.data_ptr(), d_c.data_ptr(), 0, s.cuda_stream)
        g2.capture_end()
    assert int(d_out.abs().sum().item()) == 0 and int(d_m.sum().item()) == 0      # captured, not run
    assert gpu.info("derived_bytes") == got
    g1.replay()
    g2.replay()
    torch.cuda.synchronize()
    assert (d_out.cpu().numpy().view(np.uint16) == exp).all()
    assert (d_m.cpu().numpy().view(np.uint64) == em).all() and (d_c.cpu().numpy().view(np.uint64) == ec).all()
    assert torch.cuda.mem_get_info()[0] >= free0 - (8 << 20)  # (the graphs' own bookkeeping aside)
    assert gpu.last_launch()["kernel"] == "zml_kernel_flat<6, unsigned int, 0, 0, 0, 1>"   # the count query's state machine
    gpu.close()
    cpu.close()


def test_host_staging_reserved_up_front(built_lib, golden_image):
    """"reserve_host_bases" / "reserve_host_results" / "reserve_host_reads" (round 5): the device staging of the synchronous *_host
    calls is allocated by the options, a host call within those sizes allocates nothing more (movi_index_info "host_staging_bytes"
    and the device's free memory stay where they were), its answers are the oracle's, and "release_scratch" gives it all back."""
    import torch
    import movi_amd
    from oracle.oracle import Oracle
    img = golden_image(6)
    cpu = Oracle(img)
    rng = np.random.default_rng(9800)
    reads = mutated_reads(rng, _ref(), 700, 1, 300)
    bases, offs = pack(reads)
    exp, _, _ = cpu.pml_batch(bases, offs, threads=4)
    em, ec = cpu.count_batch(bases, offs, threads=4)
    gpu = movi_amd.MoveIndex.from_image(img)
    gpu.prepare()
    assert gpu.info("host_staging_bytes") == 0
    with pytest.raises(movi_amd.MoviError):
        gpu.set_option("reserve_host_bases", -1)
    gpu.set_option("reserve_host_bases", 1 << 20)
    gpu.set_option("reserve_host_results", 1 << 20)
    gpu.set_option("reserve_host_reads", 4096)
    held = gpu.info("host_staging_bytes")
    assert held >= (1 << 20) * 3 + 4097 * 8 + 4096
    gpu.set_option("reserve_host_bases", 1 << 10)              # grow-only: a smaller request changes nothing
    assert gpu.info("host_staging_bytes") == held
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    got, _ = gpu.query_pml_packed(bases, offs)
    assert (got == exp).all()
    assert gpu.info("host_staging_bytes") == held and torch.cuda.mem_get_info()[0] >= free0 - (1 << 20)
    m, c, _ = gpu.query_count_packed(bases, offs)               # (the count call's two per-read result buffers are its own)
    assert (m == em).all() and (c == ec).all()
    gpu.set_option("release_scratch", 1)
    assert gpu.info("host_staging_bytes") == 0
    got, _ = gpu.query_pml_packed(bases, offs)                  # ... and the lazy path still serves
    assert (got == exp).all() and gpu.info("host_staging_bytes") > 0
    gpu.close()
    cpu.close()
