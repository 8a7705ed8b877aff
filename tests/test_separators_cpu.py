"""CPU suite for `movi build --separators` indexes ('%' + ACGT alphabets, explicit thresholds for the
separator rows): the constructor is pinned to the reference's index-size known answers
(tests/test_build.cpp:73-97: 948232 B regular-thresholds, 711854 B blocked-thresholds), and the
oracle's PML / count / ZML on those indexes to independent restatements on the plain BWT and text.
The reference's own PML golden for separators (reads.fasta.separators.pmls.sorted, tests/test_pml.cpp:33-35)
is not in the checkout, hence the BWT-level simulation."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, read_fastx
from oracle import build_index as B
from oracle.oracle import Oracle
from test_oracle_golden import _mutated_reads, _occurrences, _zml_brute


@pytest.fixture(scope="module")
def sep_bwt():
    recs = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))
    t = B.clean_text([s for _, s in recs], separators=True)
    return t, B.bwt_and_thresholds(t)


@pytest.fixture(scope="module")
def sep_image(sep_bwt):
    _, (bwt, thr) = sep_bwt
    cache = {}

    def get(mode):
        if mode not in cache:
            cache[mode] = B.serialize(B.build_rows(bwt, thr, mode))
        return cache[mode]
    return get


# tests/test_build.cpp:79 and :95 of the reference
@pytest.mark.parametrize("mode,size", [(6, 948232), (8, 711854)])
def test_separator_index_size_known_answers(sep_bwt, mode, size):
    _, (bwt, thr) = sep_bwt
    f = B.build_rows(bwt, thr, mode)
    assert len(B.serialize(f)) == size
    assert f["sep"] == 1 and f["alphabet"] == b"%ACGT" and f["n"] == 159843
    # row 0 (suffix "$" follows the last '%'), the '%' run inside the table and the '$' row hold an entry each
    assert len(f["sep_thr"]) == 3 and sorted(f["sep_map"]) == sorted({0, f["end_bwt_idx"]} | set(f["sep_map"]))


class BwtWalker:
    """PML restated on the plain BWT, no move rows: position p in the BWT, LF by rank, and on a
    mismatch the definition of a threshold itself -- between the previous and the next occurrence of
    the wanted character, jump down (to the next occurrence) iff p is at or past the position of the
    leftmost minimum LCP of that gap, else up; no previous occurrence: down, no next one: up."""

    def __init__(self, t, sep):
        self.sep = sep
        sa = B.suffix_array(t)
        self.bwt = t[sa - 1]
        self.lcp = B.lcp_array(t, sa)
        self.n = len(t)
        self.end = int(np.flatnonzero(self.bwt == 0)[0])
        chars = sorted(set(self.bwt.tolist()) - {0})
        self.C, self.occ, self.pos = {}, {}, {}
        tot = 1
        for c in chars:
            m = self.bwt == c
            self.C[c] = tot
            self.occ[c] = np.concatenate(([0], np.cumsum(m)))
            self.pos[c] = np.flatnonzero(m)
            tot += int(m.sum())

    def char_at(self, p):
        c = int(self.bwt[p])
        if c == 0:                                   # the '$' row's character field is 0: 'A', or '%' with separators
            return 37 if self.sep else 65
        return c

    def lf(self, p):
        c = int(self.bwt[p])
        if c == 0:
            return 0
        return self.C[c] + int(self.occ[c][p])

    def pml(self, R):
        out, ml, p = [], 0, self.n - 1
        for k, ch in enumerate(reversed(R)):
            if k:
                p = self.lf(p)
            if ch not in b"ACGT":
                ml = 0
            elif self.char_at(p) == ch:
                ml += 1
            else:
                ps = self.pos[ch]
                j = int(np.searchsorted(ps, p))      # ps[j-1] < p < ps[j]
                if j == 0:
                    p = int(ps[0])
                elif j == len(ps):
                    p = int(ps[-1])
                else:
                    up, dn = int(ps[j - 1]), int(ps[j])
                    thr = up + 1 + int(np.argmin(self.lcp[up + 1: dn + 1]))
                    p = dn if p >= thr else up
                ml = 0
            out.append(min(ml, 65535))
        return out


def test_bwt_walker_reproduces_the_pinned_oracle(golden_image):
    """The simulation is itself held to the golden-pinned no-separator oracle first."""
    recs = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))
    w = BwtWalker(B.clean_text([s for _, s in recs]), sep=0)
    o = Oracle(golden_image(6))
    rng = np.random.default_rng(5)
    reads = _mutated_reads(recs[0][1], rng, 40, 1, 300) + [s for _, s in read_fastx(os.path.join(GOLDEN, "sample.fastq"))][:6]
    for R in reads:
        assert o.pml(R).tolist() == w.pml(R), R


@pytest.mark.parametrize("mode", [6, 8])
def test_separator_pml_equals_bwt_simulation(sep_bwt, sep_image, mode):
    t, _ = sep_bwt
    w = BwtWalker(t, sep=1)
    o = Oracle(sep_image(mode))
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(6 + mode)
    reads = _mutated_reads(ref, rng, 60, 1, 300)
    reads += [s for _, s in read_fastx(os.path.join(GOLDEN, "sample.fastq"))][:8]
    # reads that run over the genome / reverse-complement junction, i.e. over a separator in the text
    L = len(ref)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    clean = bytes(t[:L])
    rc = clean.translate(comp)[::-1]
    reads += [clean[L - 40:] + rc[:40], clean[L - 5:] + b"%" + rc[:30], rc[-60:], b"%", b"A%C", b"%%ACGT"]
    for R in reads:
        assert o.pml(R).tolist() == w.pml(R), R
    # '%' in a read is illegal (check_alphabet, src/move_structure.cpp:384-388)
    assert o.pml(b"ACG%T")[1] == 0


@pytest.mark.parametrize("mode", [6, 8])
def test_separator_count_and_zml_equal_brute_force(sep_bwt, sep_image, mode):
    t, _ = sep_bwt
    T = bytes(t[:-1])
    o = Oracle(sep_image(mode))
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(31)
    reads = _mutated_reads(ref, rng, 50, 1, 120, sub=0.02, ill=0.005) + [b"A", b"ACGT", b"TTTTTTTT", b"GNAC", b"AC%GT"]
    for R in reads:
        m, c = o.count(R)
        if R[-1:] not in (b"A", b"C", b"G", b"T"):
            assert (m, c) == (0, 0)
            continue
        k = 1
        while k < len(R) and R[len(R) - k - 1:len(R) - k] in (b"A", b"C", b"G", b"T") and R[len(R) - k - 1:] in T:
            k += 1
        assert (m, c) == (k, _occurrences(T, R[len(R) - k:])), R
    for R in _mutated_reads(ref, rng, 40, 1, 200) + [b"NA", b"AN", b"ACGTNNACGT", b"AC%GT"]:
        Rz = R.replace(b"%", b"N")                  # the brute-force parse only knows ACGT as legal
        assert o.zml(R).tolist() == _zml_brute(T, Rz), R


def test_separator_modes_agree(sep_image):
    o6, o8 = Oracle(sep_image(6)), Oracle(sep_image(8))
    for _, R in read_fastx(os.path.join(GOLDEN, "sample.fastq")):
        assert (o6.pml(R) == o8.pml(R)).all()
        assert o6.count(R) == o8.count(R)


def _multi_sequence_case(rng, n_seqs):
    """Many short records -> many separator rows, and walks that keep stepping on them."""
    seqs = []
    base = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(30, 200))).astype(np.uint8))
    for _ in range(n_seqs):
        s = bytearray(base if rng.random() < 0.5 else
                      bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, 120))).astype(np.uint8)))
        for k in range(len(s)):
            if rng.random() < 0.05:
                s[k] = b"ACGT"[rng.integers(0, 4)]
        seqs.append(bytes(s))
    t = B.clean_text(seqs, separators=True)
    text = bytes(t[:-1])
    reads = []
    for _ in range(80):
        L = int(rng.integers(1, 150))
        p = int(rng.integers(0, max(1, len(text) - L)))
        r = bytearray(text[p:p + L])                # may run over '%' (illegal in a read)
        for k in range(len(r)):
            if rng.random() < 0.06:
                r[k] = b"ACGTN"[rng.integers(0, 5)]
        reads.append(bytes(r))
    return seqs, t, reads


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_separator_fuzz_many_sequences(seed):
    rng = np.random.default_rng(4000 + seed)
    seqs, t, reads = _multi_sequence_case(rng, int(rng.integers(2, 40)))
    w = BwtWalker(t, sep=1)
    T = bytes(t[:-1])
    bwt, thr = B.bwt_and_thresholds(t)
    for mode in (6, 8):
        f = B.build_rows(bwt, thr, mode)
        assert len(f["sep_thr"]) >= 2
        o = Oracle(B.serialize(f))
        for R in reads:
            assert o.pml(R).tolist() == w.pml(R), (seed, mode, R)
            m, c = o.count(R)
            if R[-1:] in (b"A", b"C", b"G", b"T"):
                k = 1
                while k < len(R) and R[len(R) - k - 1:len(R) - k] in (b"A", b"C", b"G", b"T") and R[len(R) - k - 1:] in T:
                    k += 1
                assert (m, c) == (k, _occurrences(T, R[len(R) - k:])), (seed, mode, R)
