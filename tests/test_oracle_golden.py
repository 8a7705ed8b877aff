"""CPU suite: pins the oracle (and the index constructor that feeds it) to the
golden vectors the reference's own tests hold for the PML path."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, classify_py, golden_sorted_pmls, read_fastx, stdout_line
from oracle import build_index as B
from oracle.oracle import Oracle


@pytest.fixture(scope="module")
def ref_bwt():
    recs = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))
    t = B.clean_text([s for _, s in recs])
    return B.bwt_and_thresholds(t)


# tests/test_build.cpp:37 and :53 of the reference: index sizes for tests_data/ref.fasta
@pytest.mark.parametrize("mode,size", [(6, 948119), (8, 711733)])
def test_index_size_known_answers(ref_bwt, golden_image, mode, size):
    bwt, thr = ref_bwt
    f = B.build_rows(bwt, thr, mode)
    img = B.serialize(f)
    assert len(img) == size
    assert f["r"] == 118209 and f["original_r"] == 108629 and f["n"] == 159841
    # the committed fixture is exactly what the constructor produces
    assert img == golden_image(mode)


# tests/test_pml.cpp:89-105: sample.fastq against sample.fastq.pmls.sorted (sorted multiset)
@pytest.mark.parametrize("mode", [6, 8])
def test_oracle_reproduces_golden_pmls(golden_image, mode):
    o = Oracle(golden_image(mode))
    reads = read_fastx(os.path.join(GOLDEN, "sample.fastq"))
    assert len(reads) == 25
    gold_pml, gold_ids = golden_sorted_pmls()
    mine = sorted(stdout_line(o.pml(seq)) for _, seq in reads)
    assert mine == gold_pml
    assert sorted(">" + i.decode() for i, _ in reads) == gold_ids


def test_oracle_batch_equals_single(golden_image):
    o = Oracle(golden_image(6))
    reads = [s for _, s in read_fastx(os.path.join(GOLDEN, "sample.fastq"))]
    bases = np.frombuffer(b"".join(reads), np.uint8)
    offs = np.concatenate(([0], np.cumsum([len(r) for r in reads]))).astype(np.uint64)
    out, ff, sc = o.pml_batch(bases, offs, threads=2, strands=4)
    for i, r in enumerate(reads):
        assert (out[int(offs[i]):int(offs[i + 1])] == o.pml(r)).all()


def test_modes_agree_and_illegal_chars(golden_image):
    o6, o8 = Oracle(golden_image(6)), Oracle(golden_image(8))
    rng = np.random.default_rng(7)
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    for _ in range(40):
        L = int(rng.integers(1, 400))
        s = int(rng.integers(0, len(ref) - L))
        r = bytearray(ref[s:s + L])
        for k in range(L):
            u = rng.random()
            if u < 0.03:
                r[k] = b"ACGT"[rng.integers(0, 4)]
            elif u < 0.04:
                r[k] = ord("N")
            elif u < 0.05:
                r[k] = ord("a")
        p6, p8 = o6.pml(bytes(r)), o8.pml(bytes(r))
        assert (p6 == p8).all()
        # illegal characters (N, lower case) give PML 0: check_alphabet, move_structure.cpp:383-397
        rr = np.frombuffer(bytes(r), np.uint8)[::-1]
        bad = ~np.isin(rr, np.frombuffer(b"ACGT", np.uint8))
        assert (p6[bad] == 0).all()
        assert o6.count(bytes(r)) == o8.count(bytes(r))


def test_exact_substrings_match_fully(golden_image):
    """A substring of the reference matches end to end: PML grows by one per base
    once it is positive (until a reposition), and count >= 1 with matched == len."""
    o = Oracle(golden_image(6))
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(11)
    for _ in range(30):
        L = int(rng.integers(20, 300))
        s = int(rng.integers(0, len(ref) - L))
        m, c = o.count(ref[s:s + L])
        assert m == L and c >= 1


def test_count_edge_cases(golden_image):
    o = Oracle(golden_image(6))
    assert o.count(b"N") == (0, 0)                  # move_structure_search.cpp:344-347
    assert o.count(b"ACGTN") == (0, 0)
    m, c = o.count(b"A")
    assert m == 1 and c > 0
    m, c = o.count(b"NA")
    assert m == 1                                   # stops at the illegal base
    assert o.pml(b"").size == 0


def test_add_ml_clamp():
    """include/move_query.hpp:26-38 (tests/test_basics.cpp:304-315): PML is clamped
    to 65535 while the counter keeps growing.  A 2-run text A^k makes the walk match
    forever; check on a synthetic homopolymer index built by the constructor."""
    img = B.build_index_from_seqs([b"A" * 70000], 6, rc=False)
    o = Oracle(img)
    p = o.pml(b"A" * 66000)
    assert p[0] == 1 and p[65534] == 65535 and p[65535] == 65535 and p[-1] == 65535
    assert (np.diff(p[:65535].astype(np.int64)) == 1).all()


# ---------------------------------------------------------------- count / ZML against the text
# The reference's tests hold no count or ZML vector; these pin the restated interval walkers
# (update_interval + 2 LF) to an independent brute-force substring search over the fixture text.

def _occurrences(T, P):
    n, i = 0, T.find(P)
    while i >= 0:
        n += 1
        i = T.find(P, i + 1)
    return n


def _zml_brute(T, R):
    """Greedy Ziv-Merhav parse from the right with `in T` as the only matching primitive."""
    legal = set(b"ACGT")
    out = []
    pos = len(R) - 1
    while pos >= 0 and R[pos] not in legal:
        out.append(0)
        pos -= 1
    if pos < 0:
        return out
    end, ml, alive = pos, 0, True                   # current phrase = R[pos..end]
    while pos > 0:
        if R[pos - 1] in legal and R[pos - 1:end + 1] in T:
            out.append(ml); pos -= 1; ml += 1
        else:
            out.append(ml); pos -= 1; ml = 0
            alive = False
            while R[pos] not in legal and pos > 0:
                out.append(0); pos -= 1
            if R[pos] in legal:
                end, alive = pos, True
    out.append(ml if alive else 0)
    return out


def _mutated_reads(ref, rng, n, lo, hi, sub=0.03, ill=0.01):
    reads = []
    for _ in range(n):
        L = int(rng.integers(lo, hi))
        s = int(rng.integers(0, len(ref) - L))
        r = bytearray(ref[s:s + L])
        for k in range(L):
            u = rng.random()
            if u < sub:
                r[k] = b"ACGT"[rng.integers(0, 4)]
            elif u < sub + ill:
                r[k] = b"Na"[rng.integers(0, 2)]
        reads.append(bytes(r))
    return reads


@pytest.mark.parametrize("mode", [6, 8])
def test_count_equals_brute_force(golden_image, mode):
    o = Oracle(golden_image(mode))
    recs = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))
    T = bytes(B.clean_text([s for _, s in recs])[:-1])
    rng = np.random.default_rng(21)
    reads = _mutated_reads(recs[0][1], rng, 60, 1, 120, sub=0.02, ill=0.005) + [b"A", b"ACGT", b"TTTTTTTT", b"GNAC"]
    for R in reads:
        m, c = o.count(R)
        if R[-1:] not in (b"A", b"C", b"G", b"T"):
            assert (m, c) == (0, 0)
            continue
        # longest suffix of R (ending at the last base, legal characters only) that occurs in T
        k = 1
        while k < len(R) and R[len(R) - k - 1:len(R) - k] in (b"A", b"C", b"G", b"T") and R[len(R) - k - 1:] in T:
            k += 1
        assert m == k
        assert c == _occurrences(T, R[len(R) - k:])


@pytest.mark.parametrize("mode", [6, 8])
def test_zml_equals_brute_force(golden_image, mode):
    o = Oracle(golden_image(mode))
    recs = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))
    T = bytes(B.clean_text([s for _, s in recs])[:-1])
    rng = np.random.default_rng(22)
    reads = _mutated_reads(recs[0][1], rng, 60, 1, 200)
    reads += [b"A", b"N", b"NN", b"NA", b"AN", b"NAN", b"ACGTNNACGT", b"aACGT", b"ACGTa", b"NNNNACGTACGTNNN"]
    reads += [s for _, s in read_fastx(os.path.join(GOLDEN, "sample.fastq"))][:5]
    for R in reads:
        z = o.zml(R)
        assert z.size == len(R)
        assert z.tolist() == _zml_brute(T, R), R


def test_zml_batch_and_clamp(golden_image):
    o = Oracle(golden_image(6))
    reads = [s for _, s in read_fastx(os.path.join(GOLDEN, "sample.fastq"))]
    bases = np.frombuffer(b"".join(reads), np.uint8)
    offs = np.concatenate(([0], np.cumsum([len(r) for r in reads]))).astype(np.uint64)
    out = o.zml_batch(bases, offs, threads=2)
    for i, r in enumerate(reads):
        assert (out[int(offs[i]):int(offs[i + 1])] == o.zml(r)).all()
    img = B.build_index_from_seqs([b"A" * 70000], 6, rc=False)
    z = Oracle(img).zml(b"A" * 66000)               # one phrase: 0, 1, 2, ... clamped at 65535
    assert z[0] == 0 and z[65535] == 65535 and z[-1] == 65535
    assert (np.diff(z[:65536].astype(np.int64)) == 1).all()


# tests/test_classification.cpp:54-100 of the reference: `query --pml --filter --invert --stdout` on sample.fasta,
# sorted, equals sample.fasta.pmls.filtered_notfound.sorted.  The null database is regenerated here the way
# `movi build` does (reversed random 150-bp chunks of the reference, src/utils.cpp:427-475; statistics
# src/emperical_null_database.cpp:47-92) -- the reference seeds it with time(0), so its golden holds for any draw.
@pytest.mark.parametrize("mode", [6, 8])
@pytest.mark.parametrize("seed", [1, 2])
def test_reference_filter_invert_golden(golden_image, mode, seed):
    o = Oracle(golden_image(mode))
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(seed)
    vals = []
    for _ in range(100):
        at = int(rng.integers(0, len(ref) - 150))
        vals.append(o.pml(ref[at:at + 150][::-1]))
    uniq, cnt = np.unique(np.concatenate(vals), return_counts=True)
    thr = max(int(uniq[cnt >= 5].max()), 3) + 1                     # src/classifier.cpp:32
    out = []
    for rid, seq in read_fastx(os.path.join(GOLDEN, "sample.fasta")):
        found, _, _, _ = classify_py(o.pml(seq), thr)
        if not found:                                               # --invert: output_read, src/utils.cpp:291-294
            out += [b">" + rid + b"\n", seq + b"\n"]
    assert b"".join(sorted(out)) == open(os.path.join(GOLDEN, "sample.fasta.pmls.filtered_notfound.sorted"), "rb").read()


# ---------------------------------------------------------------- sampled-thresholds (mode 7, 3-byte rows + sampled ids)
# tests/test_build.cpp:45-47 (475326 B) and :86-88 (505009 B with --separators); tests/test_pml.cpp:98-100 holds the
# sampled-thresholds index to the same golden file as the other two modes.

@pytest.fixture(scope="module")
def sampled_image(ref_bwt):
    bwt, thr = ref_bwt
    return B.serialize(B.build_rows(bwt, thr, 7))


def test_sampled_index_size_known_answers(ref_bwt, sampled_image):
    assert len(sampled_image) == 475326
    recs = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))
    t = B.clean_text([s for _, s in recs], separators=True)
    assert len(B.serialize(B.build_rows(*B.bwt_and_thresholds(t), 7))) == 505009


def test_sampled_oracle_reproduces_golden_pmls(sampled_image):
    o = Oracle(sampled_image)
    assert o.r == 118209                            # 511-base rows never split further on this text
    reads = read_fastx(os.path.join(GOLDEN, "sample.fastq"))
    gold_pml, _ = golden_sorted_pmls()
    assert sorted(stdout_line(o.pml(seq)) for _, seq in reads) == gold_pml


def test_sampled_lf_equals_regular_lf(sampled_image, golden_image):
    """get_id from the sampled checkpoints (src/move_structure.cpp:104-283) lands on the row the stored id of the
    regular index names, for every row: both tables have the same 118209 rows on this text."""
    o7, o6 = Oracle(sampled_image), Oracle(golden_image(6))
    rng = np.random.default_rng(3)
    rows = np.concatenate((np.arange(0, 200), rng.integers(0, o6.r, 4000), [o6.r - 1, o6.end_bwt_idx]))
    for i in rows.tolist():
        assert o7.lf(i, 0) == o6.lf(i, 0), i


def test_sampled_count_and_zml_equal_brute_force(sampled_image):
    o = Oracle(sampled_image)
    recs = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))
    T = bytes(B.clean_text([s for _, s in recs])[:-1])
    rng = np.random.default_rng(23)
    for R in _mutated_reads(recs[0][1], rng, 40, 1, 120, sub=0.02, ill=0.005) + [b"A", b"GNAC"]:
        m, c = o.count(R)
        if R[-1:] not in (b"A", b"C", b"G", b"T"):
            assert (m, c) == (0, 0)
            continue
        k = 1
        while k < len(R) and R[len(R) - k - 1:len(R) - k] in (b"A", b"C", b"G", b"T") and R[len(R) - k - 1:] in T:
            k += 1
        assert (m, c) == (k, _occurrences(T, R[len(R) - k:]))
    for R in _mutated_reads(recs[0][1], rng, 30, 1, 200) + [b"NA", b"ACGTNNACGT"]:
        assert o.zml(R).tolist() == _zml_brute(T, R), R


# ---------------------------------------------------------------- sampled (mode 5: the sampled layout without thresholds)
# tests/test_build.cpp:41-43 (437006 B) and :82-84 (464203 B with --separators).  PML on an index without thresholds
# repositions randomly in the reference (reposition_randomly): count and ZML are the reproducible queries.

def test_sampled_no_thresholds_known_answers_and_queries(ref_bwt):
    bwt, thr = ref_bwt
    f = B.build_rows(bwt, thr, 5)
    img = B.serialize(f)
    assert len(img) == 437006 and f["r"] == f["original_r"] == 108629      # rows = BWT runs: no threshold splits
    recs = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))
    t = B.clean_text([s for _, s in recs], separators=True)
    assert len(B.serialize(B.build_rows(*B.bwt_and_thresholds(t), 5))) == 464203
    o = Oracle(img)
    with pytest.raises(Exception):
        o.pml(b"ACGT")
    T = bytes(B.clean_text([s for _, s in recs])[:-1])
    rng = np.random.default_rng(25)
    for R in _mutated_reads(recs[0][1], rng, 40, 1, 120, sub=0.02, ill=0.005) + [b"A", b"GNAC"]:
        m, c = o.count(R)
        if R[-1:] not in (b"A", b"C", b"G", b"T"):
            assert (m, c) == (0, 0)
            continue
        k = 1
        while k < len(R) and R[len(R) - k - 1:len(R) - k] in (b"A", b"C", b"G", b"T") and R[len(R) - k - 1:] in T:
            k += 1
        assert (m, c) == (k, _occurrences(T, R[len(R) - k:]))
    for R in _mutated_reads(recs[0][1], rng, 30, 1, 200) + [b"NA", b"ACGTNNACGT"]:
        assert o.zml(R).tolist() == _zml_brute(T, R), R


# ---- `regular` (mode 3) and `blocked` (mode 2): the threshold-less siblings of modes 6 / 8 (count and ZML only:
# without thresholds the reference's PML repositions randomly).  Pinned to the reference's index-size known answers
# 871479 / 654253 B plain and 871496 / 654280 B with --separators (tests/test_build.cpp:33,49,76,92), to the stored ids
# of the KAT-pinned regular-thresholds index (same LF, rows merged where that index splits at thresholds) and to the
# brute-force count / ZML.
@pytest.mark.parametrize("mode,size,sep_size", [(3, 871479, 871496), (2, 654253, 654280)])
def test_no_threshold_modes_known_answers_and_queries(ref_bwt, golden_image, mode, size, sep_size):
    bwt, thr = ref_bwt
    f = B.build_rows(bwt, thr, mode)
    img = B.serialize(f)
    assert len(img) == size and f["r"] == f["original_r"] == 108629        # rows = BWT runs: no threshold splits
    recs = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))
    t = B.clean_text([s for _, s in recs], separators=True)
    assert len(B.serialize(B.build_rows(*B.bwt_and_thresholds(t), mode))) == sep_size
    o = Oracle(img)
    assert (o.mode, o.r) == (mode, 108629)
    with pytest.raises(Exception):
        o.pml(b"ACGT")
    # every row's LF destination = the BWT position the constructor derived (pp_id is what get_id must return)
    assert (o.get_ids().astype(np.int64) == f["pp_id"]).all()
    T = bytes(B.clean_text([s for _, s in recs])[:-1])
    rng = np.random.default_rng(30 + mode)
    for R in _mutated_reads(recs[0][1], rng, 40, 1, 120, sub=0.02, ill=0.005) + [b"A", b"GNAC"]:
        m, c = o.count(R)
        if R[-1:] not in (b"A", b"C", b"G", b"T"):
            assert (m, c) == (0, 0)
            continue
        k = 1
        while k < len(R) and R[len(R) - k - 1:len(R) - k] in (b"A", b"C", b"G", b"T") and R[len(R) - k - 1:] in T:
            k += 1
        assert (m, c) == (k, _occurrences(T, R[len(R) - k:]))
    for R in _mutated_reads(recs[0][1], rng, 30, 1, 200) + [b"NA", b"ACGTNNACGT"]:
        assert o.zml(R).tolist() == _zml_brute(T, R), R
    # the same queries on the KAT- and golden-pinned regular-thresholds index give the same answers
    o6 = Oracle(golden_image(6))
    reads = _mutated_reads(recs[0][1], rng, 50, 1, 300)
    for R in reads:
        assert o.count(R) == o6.count(R) and o.zml(R).tolist() == o6.zml(R).tolist()
