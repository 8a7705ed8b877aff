"""GPU suite (-m gpu): PML as reset masks (round 6; include/movi_hip.h "PML as reset masks").

PML[k] = reset(k) ? 0 : PML[k - 1] + 1 (/root/reference/src/read_processor.cpp:193-215, include/move_query.hpp:26-38): the walk
writes one bit per base and two expanders (device kernel, host worker threads) rebuild the u16 vector.  Everything here is held to
the oracle's vectors: the masks bit for bit (bit = PML == 0, the header's layout), the expanded vectors element for element."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from test_gpu_parity import mutated_reads, pack

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engines(built_lib, golden_image):
    import movi_amd
    from oracle.oracle import Oracle
    return {mode: (movi_amd.MoveIndex.from_image(golden_image(mode)), Oracle(golden_image(mode))) for mode in (6, 8)}


def ref_text():
    from oracle import build_index as B
    return B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]


def _pinned_copy(a):
    import movi_amd
    p = movi_amd.pinned_empty(a.size, a.dtype)
    p[:] = a
    return p


@pytest.mark.parametrize("mode", [6, 8])
def test_mask_host_and_expanders_vs_oracle(engines, mode):
    """movi_pml_mask_host's words are the oracle's (PML == 0) bits; both expanders give back the oracle's vector."""
    import torch
    from movi_amd import engine as E
    gpu, cpu = engines[mode]
    rng = np.random.default_rng(7100 + mode)
    reads = mutated_reads(rng, ref_text(), 600, 1, 900) + [b"", b"A", b"N", b"N" * 33, b"acgt" * 9, b"ACGT" * 300, b"T" * 31, b"G" * 32,
                                                           b"C" * 33, b"", b"GATTACA"] + mutated_reads(rng, ref_text(), 40, 1500, 3000)
    bases, offs = pack(reads)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=4)
    mexp, valid = E.masks_of_pml(exp, offs)
    words, st = gpu.query_pml_mask_packed(bases, offs)
    assert gpu.last_launch()["kernel"].endswith(", 2>"), gpu.last_launch()          # the walk wrote them itself (RING = 2)
    assert (words[valid] == mexp[valid]).all()
    assert (st.fast_forwards, st.scans, st.errors, st.bases) == (ff, sc, 0, bases.size)
    for th in (1, 3):
        assert (E.expand_masks_host(words, offs, threads=th) == exp).all()
    # on the device: masks -> vector
    dev = torch.device("cuda", 0)
    d_w = torch.from_numpy(words.view(np.int32).copy()).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64).copy()).to(dev)
    d_out = torch.full((bases.size,), -1, dtype=torch.int16, device=dev)
    gpu.pml_expand_device(d_w.data_ptr(), d_offs.data_ptr(), len(reads), bases.size, d_out.data_ptr())
    torch.cuda.synchronize()
    assert (d_out.cpu().numpy().view(np.uint16) == exp).all()


@pytest.mark.parametrize("mode", [6, 8])
def test_sub_batches_concatenate(engines, mode):
    """first_base: sub-batches of one read set write, word for word, what one call over the whole set writes."""
    import torch
    from movi_amd import engine as E
    gpu, cpu = engines[mode]
    rng = np.random.default_rng(7200 + mode)
    reads = mutated_reads(rng, ref_text(), 500, 1, 400)
    bases, offs = pack(reads)
    exp, _, _ = cpu.pml_batch(bases, offs, threads=4)
    mexp, valid = E.masks_of_pml(exp, offs)
    dev = torch.device("cuda", 0)
    d_bases = torch.from_numpy(bases.copy()).to(dev)
    whole = torch.zeros(mexp.size, dtype=torch.int32, device=dev)
    cuts = [0, 1, 77, 78, 300, len(reads)]
    for a, b in zip(cuts[:-1], cuts[1:]):
        b0, nb = int(offs[a]), int(offs[b] - offs[a])
        rel = (offs[a:b + 1] - offs[a]).astype(np.uint64)
        d_rel = torch.from_numpy(rel.view(np.int64).copy()).to(dev)
        nw = E.mask_words(b - a, nb, b0)
        d_w = torch.zeros(nw, dtype=torch.int32, device=dev)
        gpu.pml_mask_device(d_bases.data_ptr() + b0, d_rel.data_ptr(), b - a, nb, d_w.data_ptr(), first_base=b0)
        torch.cuda.synchronize()
        n_own = ((nb + (b0 & 31)) >> 5) + (b - a)                                 # the sub-batch's own extent (the header's rule)
        g0 = (b0 >> 5) + a
        whole[g0:g0 + n_own] = d_w[:n_own]
        # ... and the sub-batch alone expands to its slice of the vector
        d_out = torch.zeros(max(nb, 1), dtype=torch.int16, device=dev)
        gpu.pml_expand_device(d_w.data_ptr(), d_rel.data_ptr(), b - a, nb, d_out.data_ptr(), first_base=b0)
        torch.cuda.synchronize()
        assert (d_out.cpu().numpy().view(np.uint16)[:nb] == exp[b0:b0 + nb]).all(), (a, b)
    got = whole.cpu().numpy().view(np.uint32)
    assert (got[valid] == mexp[valid]).all()


def test_paths_without_a_mask_output_pack_their_vector(engines):
    """Segment-parallel long reads, the base-synchronous kernels and unstaged launches write their vector to device scratch and
    pml_to_mask_kernel packs it (one lane per read, and a wavefront per read for long reads): same words."""
    from movi_amd import engine as E
    gpu, cpu = engines[6]
    rng = np.random.default_rng(7300)
    short = mutated_reads(rng, ref_text(), 400, 1, 300)
    long_reads = mutated_reads(rng, ref_text(), 30, 2100, 5000)
    for reads, opts in ((short, {"pml_variant": 1}), (short, {"pml_variant": 0}), (short, {"stage_reads": 0}),
                        (long_reads, {"seg_len": 256, "seg_probe": 0}), (long_reads, {})):
        bases, offs = pack(reads)
        exp, ff, sc = cpu.pml_batch(bases, offs, threads=4)
        mexp, valid = E.masks_of_pml(exp, offs)
        for k, v in opts.items():
            gpu.set_option(k, v)
        try:
            words, st = gpu.query_pml_mask_packed(bases, offs)
            out = gpu.query_pml_packed(bases, offs)[0]
        finally:
            for k in opts:
                gpu.set_option(k, {"pml_variant": -1, "stage_reads": 1, "seg_len": 2048, "seg_probe": 1}[k])
        assert (words[valid] == mexp[valid]).all(), opts
        assert (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0), opts
        assert (out == exp).all()
        if "seg_len" in opts:
            assert st.segments > len(reads)
        assert (E.expand_masks_host(words, offs, threads=2) == exp).all(), opts


def test_long_reads_expand_by_wavefront(engines):
    """Mean length >= 2048: pml_expand_wave_kernel (a scan across the wavefront carries match_len from word to word), incl. runs of
    matches longer than a round of 64 words and the u16 clamp."""
    import torch
    gpu, cpu = engines[6]
    ref = ref_text()
    rng = np.random.default_rng(7400)
    reads = [ref[1000:1000 + 9000], ref[20000:20000 + 2049], ref[5:5 + 4096], ref[40000:40000 + 12345]] + mutated_reads(rng, ref, 12, 2048, 7000)
    bases, offs = pack(reads)
    exp, _, _ = cpu.pml_batch(bases, offs, threads=4)
    assert exp.max() > 2048 * 2                                                    # a run of matches that spans rounds
    words, _ = gpu.query_pml_mask_packed(bases, offs)
    dev = torch.device("cuda", 0)
    d_w = torch.from_numpy(words.view(np.int32).copy()).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64).copy()).to(dev)
    d_out = torch.full((bases.size,), -1, dtype=torch.int16, device=dev)
    gpu.pml_expand_device(d_w.data_ptr(), d_offs.data_ptr(), len(reads), bases.size, d_out.data_ptr())
    torch.cuda.synchronize()
    assert (d_out.cpu().numpy().view(np.uint16) == exp).all()


def test_u16_clamp_through_masks(built_lib):
    """MoveQuery::add_ml clamps at 65535 (include/move_query.hpp:26-38): a read of 70 000 matching bases through masks + both expanders."""
    import movi_amd
    from movi_amd import engine as E
    from oracle import build_index as B
    from oracle.oracle import Oracle
    img = B.build_index_from_seqs([b"A" * 70000 + b"C" + b"A" * 300], 6, rc=False)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    reads = [b"A" * 69000, b"A" * 66000 + b"G" + b"A" * 100, b"A" * 200 + b"C" + b"A" * 66500]
    bases, offs = pack(reads)
    exp, _, _ = cpu.pml_batch(bases, offs, threads=2)
    assert exp.max() == 65535
    words, _ = gpu.query_pml_mask_packed(bases, offs)
    assert (E.expand_masks_host(words, offs, threads=2) == exp).all()
    gpu.set_option("pml_via_mask", 1)
    gpu.set_option("host_masks", 1)
    assert (gpu.query_pml_packed(bases, offs)[0] == exp).all()                     # host path: masks down, expanded by worker threads
    gpu.close()


def test_failed_reads_report_every_base_as_reset(built_lib, golden_image):
    """A read that hit one of the reference's throws reports all-zero PMLs: all-ones masks, error byte and MOVI_ERR_INVARIANT as ever."""
    import movi_amd
    from movi_amd import engine as E
    img = bytearray(golden_image(6))
    _, _, off, _ = movi_amd.parse_index_image(bytes(img))
    rows = np.frombuffer(img, np.uint8, count=118209 * 8, offset=off).reshape(-1, 8).copy()
    rows[:, 0:4] = 0xFF                                   # every destination id >= r
    img[off: off + rows.size] = rows.tobytes()
    gpu = movi_amd.MoveIndex.from_image(bytes(img))
    reads = [b"ACGTACGT", b"A", b"", b"G" * 70, b"T", b"C" * 64]
    bases, offs = pack(reads)
    exp, est, eerr, erc = gpu.query_pml_packed(bases, offs, want_err=True)
    assert erc == -6 and list(eerr) == [1, 0, 0, 1, 0, 1]
    words, st, err, rc = gpu.query_pml_mask_packed(bases, offs, want_err=True)
    assert rc == -6 and list(err) == list(eerr) and st.errors == est.errors
    mexp, valid = E.masks_of_pml(exp, offs)
    assert (words[valid] == mexp[valid]).all()
    assert (E.expand_masks_host(words, offs) == exp).all()


@pytest.mark.parametrize("chunk_bases", [0, 1, 3000, 100_000])
def test_overlapped_host_path_through_masks(engines, chunk_bases):
    """movi_pml_host with "pml_via_mask" 1 and movi_pml_mask_host, reads page-locked: chunks in flight, each chunk's words through the
    slot's page-locked block, the vector expanded by the worker pool into a PAGEABLE array; any cut gives the oracle's answer."""
    import movi_amd
    from movi_amd import engine as E
    gpu, cpu = engines[6]
    rng = np.random.default_rng(7500)
    reads = mutated_reads(rng, ref_text(), 300, 1, 1500) + [b"", b"", b"N", b"ACGT" * 100, b""] + mutated_reads(rng, ref_text(), 200, 100, 200)
    if chunk_bases == 1:
        reads = reads[:120]
    bases, offs = pack(reads)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=4)
    mexp, valid = E.masks_of_pml(exp, offs)
    pb = _pinned_copy(bases)
    gpu.set_option("pipe_chunk_bases", chunk_bases)
    gpu.set_option("pml_via_mask", 1)
    gpu.set_option("host_masks", 1)
    try:
        for rep in range(2):
            out = np.full(bases.size, 0xABCD, np.uint16)                          # pageable: host threads write it
            got, st, err, rc = gpu.query_pml_packed(pb, offs, want_err=True, out=out)
            assert rc == 0 and (out == exp).all() and not err.any()
            assert (st.bases, st.fast_forwards, st.scans, st.errors) == (bases.size, ff, sc, 0)
            words, mst = gpu.query_pml_mask_packed(pb, offs)
            assert (words[valid] == mexp[valid]).all() and (mst.fast_forwards, mst.scans) == (ff, sc)
        out2, st2 = gpu.query_pml_packed(bases, offs)                             # pageable reads: synchronous path, still through masks
        assert (out2 == exp).all() and (st2.fast_forwards, st2.scans) == (ff, sc)
    finally:
        gpu.set_option("pipe_chunk_bases", 0)
        gpu.set_option("pml_via_mask", -1)
        gpu.set_option("host_masks", -1)


@pytest.mark.parametrize("share", [0, 40, 100])
def test_overlapped_host_path_both_ways_down(engines, share):
    """movi_pml_host, reads AND vector page-locked ("host_masks" 2 = what -1 picks for big calls on such buffers): some chunks come down as
    the vector itself by DMA, the others as reset masks that the worker pool expands -- into the same caller vector, side by side."""
    import movi_amd
    gpu, cpu = engines[6]
    rng = np.random.default_rng(7600 + share)
    reads = mutated_reads(rng, ref_text(), 400, 1, 900) + [b"", b"N", b""] + mutated_reads(rng, ref_text(), 300, 100, 200)
    bases, offs = pack(reads)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=4)
    pb = _pinned_copy(bases)
    gpu.set_option("pipe_chunk_bases", 5000)
    gpu.set_option("host_masks", 2)
    gpu.set_option("host_mask_share", share)
    try:
        for rep in range(2):
            out = movi_amd.pinned_empty(bases.size, np.uint16)
            out[:] = 0xABCD
            got, st, err, rc = gpu.query_pml_packed(pb, offs, want_err=True, out=out)
            assert rc == 0 and (out == exp).all() and not err.any()
            assert (st.bases, st.fast_forwards, st.scans, st.errors) == (bases.size, ff, sc, 0)
        out2 = np.full(bases.size, 0xABCD, np.uint16)                             # pageable vector, pageable reads: one way down (masks), synchronous
        gpu.query_pml_packed(bases, offs, out=out2)
        assert (out2 == exp).all()
    finally:
        gpu.set_option("pipe_chunk_bases", 0)
        gpu.set_option("host_masks", -1)
        gpu.set_option("host_mask_share", 70)


def test_large_batch_through_masks(built_lib):
    """1 M x 150 bp: the default policy of movi_pml_host brings masks down (>= 2^22 bases); equal to the vector path, and the device
    entry point with "pml_via_mask" 1 (mask walk + pml_expand_kernel) equal to the walk that writes the vector itself."""
    import torch
    import movi_amd
    from tools import synth
    six = synth.synth_index(2_000_000, mode=6, seed=5)
    gpu = movi_amd.MoveIndex.from_image(six.image())
    bases, offs = synth.synth_reads(six, 1_000_000, 150, seed=6, sub_rate=0.01, n_rate=0.001)
    gpu.set_option("pml_via_mask", 0)
    gpu.set_option("host_masks", 0)
    exp, est = gpu.query_pml_packed(bases, offs)
    gpu.set_option("pml_via_mask", -1)
    gpu.set_option("host_masks", -1)
    got, st = gpu.query_pml_packed(bases, offs)
    assert (got == exp).all() and (st.fast_forwards, st.scans, st.repositions) == (est.fast_forwards, est.scans, est.repositions)
    assert gpu.last_launch()["kernel"].endswith(", 2>")
    pb = movi_amd.pinned_empty(bases.size, np.uint8)
    pb[:] = bases
    got2, _ = gpu.query_pml_packed(pb, offs)
    assert (got2 == exp).all()
    dev = torch.device("cuda", 0)
    d_bases = torch.from_numpy(bases.copy()).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64).copy()).to(dev)
    d_out = torch.zeros(bases.size, dtype=torch.int16, device=dev)
    gpu.set_option("pml_via_mask", 1)
    gpu.set_option("host_masks", 1)
    gpu.pml_device(d_bases.data_ptr(), d_offs.data_ptr(), offs.size - 1, bases.size, d_out.data_ptr())
    torch.cuda.synchronize()
    assert (d_out.cpu().numpy().view(np.uint16) == exp).all()
    gpu.close()


def test_parity_files_through_masks(built_lib):
    """The PML parity files of this suite once more with MOVI_PML_VIA_MASK=1: every handle then answers movi_pml_host and
    movi_pml_device from reset masks (host worker threads / pml_expand_kernel) -- same oracle comparisons, same goldens."""
    env = dict(os.environ, MOVI_PML_VIA_MASK="1")
    # (tests/test_ahead_rows_gpu.py is about the walk's own output paths and pins them; the CLI's goldens go through masks too: the binary's
    # handles read the same variable)
    files = ["tests/test_gpu_parity.py", "tests/test_top_of_walk_gpu.py", "tests/test_device_entry_gpu.py", "tests/test_cli_gpu.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + files, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
