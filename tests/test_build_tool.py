"""CPU suite: the C++ index constructor (tools/build_index.cpp: SA-IS + Kasai + thresholds + rows)
against the KAT-pinned fixtures and against the numpy constructor of the oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

TOOL = os.path.join(ROOT, "tools", "build_index")


@pytest.fixture(scope="module")
def tool():
    src = TOOL + ".cpp"
    if not os.path.exists(TOOL) or os.path.getmtime(TOOL) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", TOOL, src])
    return TOOL


@pytest.mark.parametrize("mode,name,size", [(6, "index_regular-thresholds", 948119), (8, "index_blocked-thresholds", 711733)])
def test_reproduces_reference_known_answers(tool, tmp_path, mode, name, size):
    out = str(tmp_path / "idx")
    subprocess.check_call([tool, "fasta", os.path.join(GOLDEN, "ref.fasta"), str(mode), out], stderr=subprocess.DEVNULL)
    img = open(os.path.join(out, "index.movi"), "rb").read()
    assert len(img) == size                                   # tests/test_build.cpp:37,53 of the reference
    assert img == open(os.path.join(GOLDEN, name, "index.movi"), "rb").read()


def test_matches_numpy_constructor_on_random_multi_record_fasta(tool, tmp_path):
    from oracle import build_index as B
    rng = np.random.default_rng(4)
    fa = tmp_path / "m.fa"
    seqs = []
    with open(fa, "wb") as f:
        for i in range(5):
            base = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(500, 4000))).astype(np.uint8))
            s = bytearray(base * int(rng.integers(1, 4)))       # repeats -> long LCPs, long runs
            for k in range(0, len(s), 97):
                s[k] = b"ACGTNacgt"[rng.integers(0, 9)]        # lower case / N become 'A' (prepare_ref.cpp:39-58)
            seqs.append(bytes(s))
            f.write(b">s%d desc\n" % i)
            for j in range(0, len(s), 60):
                f.write(bytes(s[j:j + 60]) + b"\n")
    for mode in (6, 8):
        out = str(tmp_path / ("o%d" % mode))
        subprocess.check_call([tool, "fasta", str(fa), str(mode), out], stderr=subprocess.DEVNULL)
        assert open(os.path.join(out, "index.movi"), "rb").read() == B.build_index_from_seqs(seqs, mode)


# tests/test_build.cpp:79 and :95 of the reference: `movi build --separators` index sizes
@pytest.mark.parametrize("mode,size", [(6, 948232), (8, 711854)])
def test_separators_known_answers_and_numpy_agreement(tool, tmp_path, mode, size):
    from oracle import build_index as B
    out = str(tmp_path / "sep")
    subprocess.check_call([tool, "fasta", os.path.join(GOLDEN, "ref.fasta"), str(mode), out, "separators"], stderr=subprocess.DEVNULL)
    img = open(os.path.join(out, "index.movi"), "rb").read()
    assert len(img) == size
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    assert img == B.build_index_from_seqs([ref], mode, separators=True)
    # many records: many rows of the separator
    rng = np.random.default_rng(9 + mode)
    seqs = [bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, 300))).astype(np.uint8)) for _ in range(40)]
    fa = tmp_path / "many.fa"
    fa.write_bytes(b"".join(b">r%d\n%s\n" % (i, s) for i, s in enumerate(seqs)))
    out2 = str(tmp_path / "sep2")
    subprocess.check_call([tool, "fasta", str(fa), str(mode), out2, "separators"], stderr=subprocess.DEVNULL)
    assert open(os.path.join(out2, "index.movi"), "rb").read() == B.build_index_from_seqs(seqs, mode, separators=True)


# tests/test_build.cpp:45-47 and :86-88 of the reference: sampled-thresholds index sizes
@pytest.mark.parametrize("separators,size", [(False, 475326), (True, 505009)])
def test_sampled_thresholds_known_answers_and_numpy_agreement(tool, tmp_path, separators, size):
    from oracle import build_index as B
    out = str(tmp_path / "m7")
    subprocess.check_call([tool, "fasta", os.path.join(GOLDEN, "ref.fasta"), "7", out] + (["separators"] if separators else []),
                          stderr=subprocess.DEVNULL)
    img = open(os.path.join(out, "index.movi"), "rb").read()
    assert len(img) == size
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    assert img == B.build_index_from_seqs([ref], 7, separators=separators)


# tests/test_build.cpp:41-43 and :82-84 of the reference: sampled (no thresholds) index sizes
@pytest.mark.parametrize("separators,size", [(False, 437006), (True, 464203)])
def test_sampled_known_answers_and_numpy_agreement(tool, tmp_path, separators, size):
    from oracle import build_index as B
    out = str(tmp_path / "m5")
    subprocess.check_call([tool, "fasta", os.path.join(GOLDEN, "ref.fasta"), "5", out] + (["separators"] if separators else []),
                          stderr=subprocess.DEVNULL)
    img = open(os.path.join(out, "index.movi"), "rb").read()
    assert len(img) == size
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    assert img == B.build_index_from_seqs([ref], 5, separators=separators)


# tests/test_build.cpp:33,49 and :76,92 of the reference: regular / blocked (no thresholds) index sizes
@pytest.mark.parametrize("mode,separators,size", [(3, False, 871479), (3, True, 871496), (2, False, 654253), (2, True, 654280)])
def test_regular_and_blocked_known_answers_and_numpy_agreement(tool, tmp_path, mode, separators, size):
    from oracle import build_index as B
    out = str(tmp_path / "m")
    subprocess.check_call([tool, "fasta", os.path.join(GOLDEN, "ref.fasta"), str(mode), out] + (["separators"] if separators else []),
                          stderr=subprocess.DEVNULL)
    img = open(os.path.join(out, "index.movi"), "rb").read()
    assert len(img) == size
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    assert img == B.build_index_from_seqs([ref], mode, separators=separators)


def test_pangenome_mode_is_queryable(tool, tmp_path):
    """Synthetic pangenome: substrings of the text are found end to end by the oracle's count query."""
    from oracle.oracle import Oracle
    out = str(tmp_path / "pg")
    subprocess.check_call([tool, "pangenome", "20000", "8", "0.005", "3", "6", out, "300", "100", "0"], stderr=subprocess.DEVNULL)
    o = Oracle(open(os.path.join(out, "index.movi"), "rb").read())
    reads = np.fromfile(os.path.join(out, "reads.bin"), np.uint8).reshape(300, 100)
    for r in reads[:100]:
        if (r != ord("N")).all():
            m, c = o.count(bytes(r))
            assert m == 100 and c >= 1
