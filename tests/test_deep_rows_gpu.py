"""GPU suite (-m gpu): DEEP ROWS ("deep_rows" option; round 6) -- the PML walk's table copy in which every row carries what the walk
reads at its LF target AND at that row's LF target, 21.33 bytes per row, windows of three rows: up to three bases per gather
(reference semantics of each ridden step: /root/reference/src/read_processor.cpp:188-238 match branch + LF_move,
src/move_structure.cpp:59-87).  Same automaton as the walk on the plain and the look-ahead rows: PMLs, error bytes, bins, reset masks
and the fast-forward / scan / reposition counters must equal the oracle's -- on every index type the PML walk serves, with and without the
top-of-walk table (every walk starts in the table's LAST, padded window), capped big batches and small ones, reads that roll through
the staged stretch, illegal bases, separators, corrupt rows, tiny tables with awkward structure, segments of long reads."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, classify_py, vector_kernel
from test_ahead_rows_gpu import _big_batch
from test_gpu_parity import check_segmented, mutated_reads, pack
from test_top_of_walk_gpu import _ref

pytestmark = pytest.mark.gpu


def _deep(gpu, on=1):
    gpu.set_option("deep_rows", on)


@pytest.mark.parametrize("mode", [6, 8, 7])
def test_deep_rows_vs_oracle(built_lib, golden_image, mode):
    import movi_amd
    from movi_amd import engine as E
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = _ref()
    img = golden_image(mode) if mode != 7 else B.build_index_from_seqs([ref], 7)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    bases, offs = _big_batch(ref, np.random.default_rng(9600 + mode))
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=8)
    _deep(gpu)
    assert gpu.info("deep_rows_bytes") == ((gpu.desc.r + 2) // 3) * 64
    for K in (0, 12):
        gpu.set_option("kmer_k", K)
        for hints in (1, 0):
            gpu.set_option("repo_hints", hints)
            gpu.set_option("pml_via_mask", hints - 1)              # (the vector itself / the default: masks down, expanded on the host)
            gpu.set_option("host_masks", hints - 1)
            out, st = gpu.query_pml_packed(bases, offs)
            li = gpu.last_launch()
            # (a host call of this size brings reset masks down: RING = 2; "pml_via_mask" 0 below: the vector itself)
            assert li["ahead"] == 2 and li["kernel"] in ("pml_kernel_flatp<6, unsigned int, 0, 0, 0, 1, 2, 0, 0>", "pml_kernel_flatp<6, unsigned int, 0, 0, 0, 1, 2, 0, 2>"), li
            assert (out == exp).all(), (K, hints)
            assert (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0), (K, hints)
        gpu.set_option("repo_hints", 1)
        # fused bins (CLS 1 / 2) and reset masks (RING 2) on the same copy
        a, b, s = gpu.classify_packed(bases, offs, 40, 4)
        for i in (0, 5, 77, 1000, 99_999, 250_000):
            if offs[i + 1] > offs[i]:
                e = classify_py(exp[int(offs[i]):int(offs[i + 1])], 4, 40)
                assert (a[i], b[i]) == (e[2], e[3]) and s[i] == round(e[1] * (e[2] + e[3])), (K, i)
        words, mst = gpu.query_pml_mask_packed(bases, offs)
        assert gpu.last_launch()["kernel"] == "pml_kernel_flatp<6, unsigned int, 0, 0, 0, 1, 2, 0, 2>"
        mexp, valid = E.masks_of_pml(exp, offs)
        assert (words[valid] == mexp[valid]).all() and (mst.fast_forwards, mst.scans) == (ff, sc)
    # "deep" 0: the copy is ignored (A/B): the look-ahead rows the first query built beside it
    gpu.set_option("deep", 0)
    out, st = gpu.query_pml_packed(bases, offs)
    assert gpu.last_launch()["ahead"] == 1 and (out == exp).all() and (st.fast_forwards, st.scans) == (ff, sc)
    gpu.set_option("deep", -1)
    # small batches (uncapped launches, staged stretch of 336) and reads that roll
    rng = np.random.default_rng(9700 + mode)
    reads = mutated_reads(rng, ref, 500, 1, 1200) + [b"", b"A", b"N", b"NN", b"ACG", b"acgt" * 5, ref[:3], ref[7:9], b"T" * 400]
    sb, so = pack(reads)
    sexp, sff, ssc = cpu.pml_batch(sb, so, threads=4)
    out, st = gpu.query_pml_packed(sb, so)
    assert gpu.last_launch()["ahead"] == 2 and (out == sexp).all() and (st.fast_forwards, st.scans, st.errors) == (sff, ssc, 0)
    gpu.set_option("deep", 1)                                    # (segments are long reads: they walk on the deep rows on request only)
    check_segmented(gpu, sb, so, sexp, sff, ssc, "deep")
    assert gpu.last_launch()["ahead"] == 2                       # K1 ran on the deep rows
    gpu.set_option("deep", -1)
    # "ahead_rows" is a statement about what the walk runs on: it frees the deep rows
    gpu.set_option("ahead_rows", 1)
    assert gpu.info("deep_rows_bytes") == 0
    out, st = gpu.query_pml_packed(sb, so)
    assert gpu.last_launch()["ahead"] == 1 and (out == sexp).all()
    gpu.close()


def test_deep_rows_default_policy(built_lib, golden_image):
    """Tables of up to 50 M rows of real text get the deep rows by themselves, beside the look-ahead rows (movi_index_prepare builds both):
    batches of short reads walk on the one, long reads on the other; a uniformly random run sequence gets none."""
    import movi_amd
    from tools import synth
    gpu = movi_amd.MoveIndex.from_image(golden_image(6))
    gpu.prepare(gpu.PREPARE_PML)
    assert gpu.info("deep_rows_bytes") > 0 and gpu.info("ahead_rows_bytes") > 0 and gpu.info("ahead_no_ff") >= 0.67
    ref = _ref()
    reads = [ref[1000 * i: 1000 * i + 160] for i in range(70)]
    bases, offs = pack(reads)
    gpu.query_pml_packed(bases, offs)
    assert gpu.last_launch()["ahead"] == 2
    long_reads = [ref[3000 * i: 3000 * i + 2000] for i in range(30)]     # mean length >= 1024: the look-ahead rows
    lb, lo = pack(long_reads)
    gpu.query_pml_packed(lb, lo)
    assert gpu.last_launch()["ahead"] == 1
    gpu.set_option("idx64", 1)                                     # (test hook: the 64-bit instantiations have no deep form)
    gpu.query_pml_packed(bases, offs)
    assert gpu.last_launch()["ahead"] == 1 and gpu.last_launch()["idx64"] == 1
    gpu.close()
    six = synth.synth_index(300_000, mode=6, seed=4)
    rnd = movi_amd.MoveIndex.from_image(six.image())
    rnd.prepare(rnd.PREPARE_PML)
    assert rnd.info("deep_rows_bytes") == 0 and rnd.info("ahead_rows_bytes") > 0 and 0 < rnd.info("ahead_no_ff") < 0.67
    rnd.close()


def test_deep_rows_separators_and_corrupt_rows(built_lib, golden_image):
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = _ref()
    img = B.build_index_from_seqs([ref[:60000], ref[60000:]], 6, separators=True)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    _deep(gpu)
    rng = np.random.default_rng(9800)
    reads = mutated_reads(rng, ref, 600, 1, 900) + [ref[59990:60010], b"%", b"A%C"]
    bases, offs = pack(reads)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=4)
    out, st = gpu.query_pml_packed(bases, offs)
    assert vector_kernel(gpu.last_launch()["kernel"]) == "pml_kernel_flatp<6, unsigned int, 0, 1, 0, 1, 2, 0, 0>"
    assert (out == exp).all() and (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0)
    gpu.close()
    # every destination id >= r: the reference throws in LF_move (src/move_structure.cpp:63-65); flagged exactly as on the other layouts
    img = bytearray(golden_image(6))
    _, _, off, _ = movi_amd.parse_index_image(bytes(img))
    rows = np.frombuffer(img, np.uint8, count=118209 * 8, offset=off).reshape(-1, 8).copy()
    rows[:, 0:4] = 0xFF
    img[off: off + rows.size] = rows.tobytes()
    bad = movi_amd.MoveIndex.from_image(bytes(img))
    b2, o2 = pack([b"ACGTACGT", b"A", b"", b"GG", b"T"] * 20)
    bad.set_option("ahead_rows", 0)
    exp, est, eerr, erc = bad.query_pml_packed(b2, o2, want_err=True)
    _deep(bad)
    out, st, err, rc = bad.query_pml_packed(b2, o2, want_err=True)
    assert bad.last_launch()["ahead"] == 2
    assert rc == erc == -6 and list(err) == list(eerr) and st.errors == est.errors and (out == exp).all()
    bad.close()
    # ... and a table in which only SOME ids are off (entries of depth one and two become invalid at different rows)
    img = bytearray(golden_image(6))
    rows = np.frombuffer(img, np.uint8, count=118209 * 8, offset=off).reshape(-1, 8).copy()
    rows[::97, 0:4] = 0xFF
    img[off: off + rows.size] = rows.tobytes()
    bad = movi_amd.MoveIndex.from_image(bytes(img))
    reads = mutated_reads(np.random.default_rng(9801), ref, 800, 20, 300)
    b3, o3 = pack(reads)
    bad.set_option("ahead_rows", 0)
    exp, est, eerr, erc = bad.query_pml_packed(b3, o3, want_err=True)
    assert est.errors > 0
    _deep(bad)
    out, st, err, rc = bad.query_pml_packed(b3, o3, want_err=True)
    assert rc == erc and list(err) == list(eerr) and st.errors == est.errors and (out == exp).all()
    assert (st.fast_forwards, st.scans) == (est.fast_forwards, est.scans)
    bad.close()


@pytest.mark.parametrize("alphabet", [b"ACGT", b"ACG", b"AT", b"C"])
def test_deep_rows_fuzz_small_indexes(built_lib, tmp_path, alphabet):
    """Tiny indexes with awkward structure (reduced alphabets, long runs split at MAX_RUN_LENGTH, the terminator row in odd places, row
    counts of every residue modulo 3: the last window's padding) through the deep rows, against the oracle."""
    import movi_amd
    from oracle.oracle import Oracle
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "build_index")
    if not os.path.exists(tool):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", tool, tool + ".cpp"])
    rng = np.random.default_rng(len(alphabet) * 77 + alphabet[0])
    seen_mod = set()
    for trial in range(8):
        recs = []
        for _ in range(int(rng.integers(1, 4))):
            unit = bytes(rng.choice(list(alphabet), size=int(rng.integers(5, 400))).astype(np.uint8))
            recs.append(unit * int(rng.integers(1, 6)) + bytes([alphabet[0]]) * int(rng.integers(0, 3000)))
        fa = tmp_path / ("f%d.fa" % trial)
        fa.write_bytes(b"".join(b">s%d\n%s\n" % (i, s) for i, s in enumerate(recs)))
        text = b"".join(recs)
        reads = []
        for _ in range(150):
            L = int(rng.integers(1, 300))
            p = int(rng.integers(0, max(1, len(text) - L)))
            r = bytearray(text[p:p + L])
            for k in range(len(r)):
                if rng.random() < 0.05:
                    r[k] = b"ACGTN"[rng.integers(0, 5)]
            reads.append(bytes(r))
        bases, offs = pack(reads)
        for mode in (6, 8):
            out_dir = str(tmp_path / ("i%d_%d" % (trial, mode)))
            subprocess.check_call([tool, "fasta", str(fa), str(mode), out_dir], stderr=subprocess.DEVNULL)
            img = open(os.path.join(out_dir, "index.movi"), "rb").read()
            gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
            if gpu.desc.r < 8:
                gpu.close()
                continue
            seen_mod.add(gpu.desc.r % 3)
            _deep(gpu)
            exp, ff, sc = cpu.pml_batch(bases, offs, threads=2)
            for K in (0, 12) if alphabet == b"ACGT" else (0,):
                gpu.set_option("kmer_k", K)
                out, st = gpu.query_pml_packed(bases, offs)
                assert gpu.last_launch()["ahead"] == 2
                assert (out == exp).all(), (alphabet, trial, mode, K)
                assert (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0), (alphabet, trial, mode, K)
            gpu.set_option("deep", 1)
            check_segmented(gpu, bases, offs, exp, ff, sc, (alphabet, trial, mode))
            gpu.close()
    assert len(seen_mod) >= 2
