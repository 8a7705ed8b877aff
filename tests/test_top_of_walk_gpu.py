"""GPU suite (-m gpu): the top-of-walk table ("kmer_k" option) -- the first K bases of every read (segment) served by one
table lookup instead of K row gathers (the analogue of the reference's ftab, src/move_structure_search.cpp:66-167).
Answers, error bytes and the fast-forward / scan / reposition counters must be those of the oracle for every K, on every
index type, for reads shorter than, as long as and longer than K, with illegal bases inside and outside the K-mer, with
fused classification bins and on the segment-parallel path."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_gpu_parity import mutated_reads, pack

pytestmark = pytest.mark.gpu


def _ref():
    from oracle import build_index as B
    return B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]


def _edge_reads(ref, rng, K):
    reads = mutated_reads(rng, ref, 300, 1, 400)
    base = bytes(ref[5000:5400])
    for L in (0, 1, K - 1, K, K + 1, K + 2, 15, 16, 17, 31, 32, 33):
        if L >= 0:
            reads.append(base[:L])
    for pos in (1, K - 1, K, K + 1, 7, 8, 9):                    # an N / a lowercase base at distance `pos` from the read's end
        for bad in (b"N", b"a"):
            r = bytearray(base[:60])
            if 0 < pos <= len(r):
                r[len(r) - pos: len(r) - pos + 1] = bad
            reads.append(bytes(r))
    reads += [b"ACGT" * 30, b"T" * 50, b"GGGGGGGGGGGGGGGGGGGGGGGGC"]
    return reads


@pytest.mark.parametrize("mode", [6, 8, 7])
@pytest.mark.parametrize("K", [1, 3, 8, 11, 12])
def test_top_of_walk_vs_oracle(built_lib, golden_image, mode, K):
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = _ref()
    img = golden_image(mode) if mode != 7 else B.build_index_from_seqs([ref], 7)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    reads = _edge_reads(ref, np.random.default_rng(8800 + K), K)
    bases, offs = pack(reads)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=4)
    gpu.set_option("kmer_k", 0)                                   # (left alone, the first PML query builds the K = 12 table)
    base_out, base_st = gpu.query_pml_packed(bases, offs)
    assert (base_out == exp).all()
    gpu.set_option("kmer_k", K)
    for variant in (-1, 1):                                       # the default walk and the base-synchronous kernel (no table there)
        gpu.set_option("pml_variant", variant)
        out, st = gpu.query_pml_packed(bases, offs)
        assert (out == exp).all(), (mode, K, variant)
        assert (st.fast_forwards, st.scans, st.repositions, st.errors) == (ff, sc, base_st.repositions, 0), (mode, K, variant)
        if variant == -1:
            # fewer iterations than without the table (the point of it; K = 1 only saves the start row's own step)
            assert K < 8 or st.lane_steps < base_st.lane_steps
    gpu.set_option("pml_variant", -1)
    # fused bins, with and without the PML vector
    exp_bins = None
    gpu.set_option("kmer_k", 0)
    exp_bins = gpu.classify_packed(bases, offs, 40, 4)
    gpu.set_option("kmer_k", K)
    got = gpu.classify_packed(bases, offs, 40, 4)
    assert all((x == y).all() for x, y in zip(got, exp_bins))
    # segment-parallel: every segment starts from the state every read starts in, so the table serves segments too
    long_reads = mutated_reads(np.random.default_rng(8900 + K), ref, 40, 700, 3000)
    lb, lo = pack(long_reads)
    lexp, lff, lsc = cpu.pml_batch(lb, lo, threads=4)
    gpu.set_option("seg_len", 64)
    gpu.set_option("seg_probe", 0)
    lout, lst = gpu.query_pml_packed(lb, lo)
    assert lst.segments > len(long_reads)
    assert (lout == lexp).all() and (lst.fast_forwards, lst.scans, lst.errors) == (lff, lsc, 0)
    # the table goes away again
    gpu.set_option("kmer_k", 0)
    gpu.set_option("seg_len", 2048)
    gpu.set_option("seg_probe", 1)
    out, st = gpu.query_pml_packed(bases, offs)
    assert (out == exp).all() and st.lane_steps == base_st.lane_steps
    gpu.close()


def test_top_of_walk_on_a_separators_index(built_lib):
    """`movi build --separators`: codes 1..4, code 0 = '%' (illegal in a read); the table is indexed by code - 1."""
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = _ref()
    seqs = [ref[:40000], ref[40000:90000], ref[90000:]]
    img = B.build_index_from_seqs(seqs, 6, separators=True)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    reads = _edge_reads(ref, np.random.default_rng(8700), 9) + [bytes(ref[39950:40050]), b"ACG%TACGTACGTACGT", b"ACGTACGTACGTAC%T"]
    bases, offs = pack(reads)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=4)
    gpu.set_option("kmer_k", 9)
    out, st = gpu.query_pml_packed(bases, offs)
    assert (out == exp).all() and (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0)
    gpu.close()


def test_top_of_walk_refuses_what_it_cannot_serve(built_lib):
    import movi_amd
    from oracle import build_index as B
    img = B.build_index_from_seqs([b"ATTATAATTTATATAATATTTAATAATTATATTTAAT" * 20], 6)    # A and T only (its reverse complement too): 2 symbols
    gpu = movi_amd.MoveIndex.from_image(img)
    with pytest.raises(movi_amd.MoviError):
        gpu.set_option("kmer_k", 8)
    with pytest.raises(movi_amd.MoviError):
        gpu.set_option("kmer_k", 13)
    out, _ = gpu.query_pml_packed(*pack([b"ATTATAATTT"]))
    assert out.size == 10
    gpu.close()


def test_top_of_walk_with_corrupt_rows(built_lib, golden_image):
    """K-mers whose walk runs into one of the reference's throws have no entry: those reads take the ordinary walk and
    report the error exactly as without the table."""
    import movi_amd
    img = bytearray(golden_image(6))
    _, _, off, _ = movi_amd.parse_index_image(bytes(img))
    rows = np.frombuffer(img, np.uint8, count=118209 * 8, offset=off).reshape(-1, 8).copy()
    rng = np.random.default_rng(8600)
    hit = rng.choice(118209, 30000, replace=False)
    rows[hit, 0:4] = 0xFF                                  # a quarter of the rows point past the table
    img[off: off + rows.size] = rows.tobytes()
    gpu = movi_amd.MoveIndex.from_image(bytes(img))
    reads = mutated_reads(rng, _ref(), 400, 5, 300)
    bases, offs = pack(reads)
    gpu.set_option("kmer_k", 0)
    e_out, e_st, e_err, e_rc = gpu.query_pml_packed(bases, offs, want_err=True)
    assert e_rc == -6 and e_st.errors > 50
    gpu.set_option("kmer_k", 6)
    out, st, err, rc = gpu.query_pml_packed(bases, offs, want_err=True)
    assert rc == -6 and (out == e_out).all() and (err == e_err).all() and st.errors == e_st.errors
    gpu.close()


@pytest.mark.parametrize("mode", [6, 8])
def test_reads_staged_through_lds_vs_oracle(built_lib, golden_image, mode):
    """Big batches (more reads than ~18 wavefronts per CU) run capped at 7 wavefronts per CU, and the LDS that enforces the
    cap holds the reads: every lane copies the next 336 bases of its read (what the padding holds) into LDS and takes every
    base from there; a wavefront with a longer read stages again, every lane from where it stands, whenever one of its lanes
    leaves its stretch.  Both kinds in one launch, every length from 0 to 336 and beyond, with and without the top-of-walk
    table, fused bins included: PMLs, error bytes and counters equal the oracle's and those of the unstaged launch."""
    import movi_amd
    from oracle.oracle import Oracle
    img = golden_image(mode)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    ref = _ref()
    rng = np.random.default_rng(8500 + mode)
    n = 300_000                                           # > 256 CUs x 64 lanes x 18 wavefronts
    CAP = 336                                             # bases per lane the default cap's LDS padding holds (7 wavefronts per CU)
    lens = rng.integers(0, CAP + 1, n).astype(np.uint64)
    lens[rng.choice(n, 40, replace=False)] = rng.integers(CAP + 1, 2000, 40)   # a few wavefronts that roll through their staged stretch
    lens[:64] = CAP                                      # one wavefront of reads that fill the staging area exactly
    lens[64:128] = 0
    starts = rng.integers(0, len(ref) - 2000, n)
    offs = np.zeros(n + 1, np.uint64)
    np.cumsum(lens, out=offs[1:])
    refa = np.frombuffer(ref, np.uint8)
    idx = np.repeat(starts.astype(np.int64) - offs[:-1].astype(np.int64), lens.astype(np.int64)) + np.arange(int(offs[-1]), dtype=np.int64)
    bases = refa[idx].copy()
    mut = rng.random(bases.size)
    bases[mut < 0.02] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int((mut < 0.02).sum()))]
    bases[(mut >= 0.02) & (mut < 0.023)] = ord("N")
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=8)
    gpu.set_option("ahead_rows", 0)                       # (tests/test_ahead_rows_gpu.py; with them the cap is 9 and the stretch 256)
    gpu.set_option("stage_reads", 0)
    gpu.set_option("kmer_k", 0)
    out0, st0 = gpu.query_pml_packed(bases, offs)
    assert gpu.last_launch()["staged"] == 0 and gpu.last_launch()["waves_per_cu"] == 7
    assert (out0 == exp).all() and (st0.fast_forwards, st0.scans, st0.errors) == (ff, sc, 0)
    bins0 = gpu.classify_packed(bases, offs, 40, 4)
    gpu.set_option("stage_reads", 1)
    for K in (0, 10):
        gpu.set_option("kmer_k", K)
        for variant in (-1, 14):
            gpu.set_option("pml_variant", variant)
            out, st = gpu.query_pml_packed(bases, offs)
            assert gpu.last_launch()["staged"] == CAP, (K, variant)
            assert gpu.last_launch()["ahead"] == 0
            assert (out == exp).all(), (K, variant)
            assert (st.fast_forwards, st.scans, st.repositions, st.errors) == (ff, sc, st0.repositions, 0), (K, variant)
        gpu.set_option("pml_variant", -1)
        bins = gpu.classify_packed(bases, offs, 40, 4)
        assert all((x == y).all() for x, y in zip(bins, bins0)), K
    gpu.close()


def test_top_of_walk_table_is_the_default(built_lib, golden_image):
    """Left alone, the first PML query on a DNA *-thresholds index builds the K = 12 table: fewer iterations, same answers;
    count and ZML queries never build it."""
    import movi_amd
    from oracle.oracle import Oracle
    img = golden_image(6)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    reads = mutated_reads(np.random.default_rng(8400), _ref(), 500, 30, 300)
    bases, offs = pack(reads)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=4)
    out, st = gpu.query_pml_packed(bases, offs)
    assert (out == exp).all() and (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0)
    gpu.set_option("kmer_k", 0)
    out0, st0 = gpu.query_pml_packed(bases, offs)
    assert (out0 == exp).all() and st.lane_steps < st0.lane_steps - 8 * len(reads)
    gpu.close()


@pytest.mark.parametrize("mode", [6, 8, 3, 2])
@pytest.mark.parametrize("K", [1, 5, 12])
def test_count_interval_table_vs_oracle(built_lib, golden_image, mode, K):
    """The count query's interval table ("ftab_k"; the reference's ftab, src/move_structure_search.cpp:66-167): the
    backward-search interval after the last K bases of a read by one lookup.  matched / count and the fast-forward / scan
    counters equal the oracle's for every K, with and without thresholds, for reads shorter than K, K-mers that do not
    occur and illegal bases inside the K-mer."""
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = _ref()
    img = golden_image(mode) if mode in (6, 8) else B.build_index_from_seqs([ref], mode)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    rng = np.random.default_rng(8300 + 10 * mode + K)
    reads = _edge_reads(ref, rng, K) + [bytes(rng.choice(list(b"ACGT"), size=40).astype(np.uint8)) for _ in range(200)]   # random 40-mers: mostly absent
    bases, offs = pack(reads)
    em, ec = cpu.count_batch(bases, offs, threads=4)
    gpu.set_option("ftab_k", 0)
    m0, c0, st0 = gpu.query_count_packed(bases, offs)
    assert (m0 == em).all() and (c0 == ec).all()
    gpu.set_option("ftab_k", K)
    m, c, st = gpu.query_count_packed(bases, offs)
    assert (m == em).all() and (c == ec).all(), (mode, K)
    assert (st.fast_forwards, st.scans, st.errors) == (st0.fast_forwards, st0.scans, 0), (mode, K)
    gpu.close()


def test_count_interval_table_is_the_default_and_optional(built_lib):
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = _ref()
    img = B.build_index_from_seqs([ref[:50000], ref[50000:]], 6, separators=True)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    reads = mutated_reads(np.random.default_rng(8200), ref, 400, 5, 300) + [b"ACGT%ACGTACGTACGTACGT", bytes(ref[49990:50010])]
    bases, offs = pack(reads)
    em, ec = cpu.count_batch(bases, offs, threads=4)
    m, c, _ = gpu.query_count_packed(bases, offs)           # left alone: the first count query builds the K = 12 table
    assert (m == em).all() and (c == ec).all()
    small = movi_amd.MoveIndex.from_image(B.build_index_from_seqs([b"ATTATAATTTATATAATATTTAATAATTATATTTAAT" * 20], 6))
    with pytest.raises(movi_amd.MoviError):
        small.set_option("ftab_k", 8)                       # two-symbol alphabet: no table
    assert small.query_count([b"ATTATA"])[0][0] == 6
    small.close()
    gpu.close()
