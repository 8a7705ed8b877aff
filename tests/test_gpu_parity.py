"""GPU suite (-m gpu): the HIP path, called through the C-ABI, against the oracle
and the committed golden fixtures.  Bit-exact: everything here is integer work."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_sorted_pmls, read_fastx, stdout_line

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engines(built_lib, golden_image):
    import movi_amd
    from oracle.oracle import Oracle
    out = {}
    for mode in (6, 8):
        img = golden_image(mode)
        out[mode] = (movi_amd.MoveIndex.from_image(img), Oracle(img))
    return out


def pack(reads):
    lens = [len(r) for r in reads]
    offs = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
    return np.frombuffer(b"".join(reads), np.uint8), offs


# the reference's own golden file (tests/test_pml.cpp:89-105)
@pytest.mark.parametrize("mode", [6, 8])
def test_golden_sample_fastq(engines, mode):
    gpu, _ = engines[mode]
    reads = [s for _, s in read_fastx(os.path.join(GOLDEN, "sample.fastq"))]
    pm = gpu.query_pml(reads)
    gold_pml, _ = golden_sorted_pmls()
    assert sorted(stdout_line(p) for p in pm) == gold_pml


def test_index_load_from_directory(built_lib):
    import movi_amd
    ix = movi_amd.MoveIndex.load(os.path.join(GOLDEN, "index_regular-thresholds"))
    assert ix.desc.r == 118209
    reads = [s for _, s in read_fastx(os.path.join(GOLDEN, "sample.fasta"))]
    gold_pml, _ = golden_sorted_pmls()
    assert sorted(stdout_line(p) for p in ix.query_pml(reads)) == gold_pml
    with pytest.raises(movi_amd.MoviError) as e:
        movi_amd.MoveIndex.load("/nonexistent/dir")
    assert e.value.code == -3


def check_segmented(gpu, bases, offs, exp, ff, sc, tag=None):
    """The same batch through the segment-parallel path (32-base segments, whatever the probe would say): same PMLs, same counters."""
    gpu.set_option("pml_variant", -1)
    gpu.set_option("seg_len", 32)
    gpu.set_option("seg_probe", 0)
    try:
        out, st = gpu.query_pml_packed(bases, offs)
        if bases.size >= 64 * (offs.size - 1):
            assert st.segments > 0, tag
        assert (out == exp).all(), tag
        assert (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0), (tag, st.segments, st.rewalked)
    finally:
        gpu.set_option("seg_len", 2048)
        gpu.set_option("seg_probe", 1)


def mutated_reads(rng, ref, n, lo, hi):
    reads = []
    for _ in range(n):
        L = int(rng.integers(lo, hi))
        s = int(rng.integers(0, len(ref) - L))
        r = bytearray(ref[s:s + L])
        for k in range(L):
            u = rng.random()
            if u < 0.02:
                r[k] = b"ACGT"[rng.integers(0, 4)]
            elif u < 0.025:
                r[k] = ord("N")
            elif u < 0.03:
                r[k] = ord("acgt"[rng.integers(0, 4)])
        reads.append(bytes(r))
    return reads


@pytest.mark.parametrize("mode", [6, 8])
def test_pml_ragged_reads_vs_oracle(engines, mode):
    from oracle import build_index as B
    gpu, cpu = engines[mode]
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(100 + mode)
    reads = mutated_reads(rng, ref, 700, 1, 2500)
    # edge cases the reference's tests and parser allow
    reads += [b"", b"A", b"N", b"NNNNNNNN", b"acgtacgt", b"ACGT" * 300, b"T" * 77, b"", b"GATTACA"]
    bases, offs = pack(reads)
    out, st = gpu.query_pml_packed(bases, offs)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=4)
    assert (out == exp).all()
    assert (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0)
    assert st.bases == bases.size


@pytest.mark.parametrize("mode", [6, 8])
@pytest.mark.parametrize("variant", [0, 1, 14])
def test_pml_kernel_variants_vs_oracle(engines, mode, variant):
    """Every selectable kernel variant is held to the same bit-exact bar, including
    reads whose length is not a multiple of the 8-step packing and unaligned offsets."""
    from oracle import build_index as B
    gpu, cpu = engines[mode]
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(300 + mode)
    reads = mutated_reads(rng, ref, 300, 1, 70) + mutated_reads(rng, ref, 300, 100, 200)
    reads += [b"", b"A", b"AC", b"ACGTACG", b"ACGTACGT", b"ACGTACGTA", b"N" * 9, b"T" * 15, b"G" * 16, b"C" * 17]
    bases, offs = pack(reads)
    gpu.set_option("pml_variant", variant)
    try:
        out, st = gpu.query_pml_packed(bases, offs)
    finally:
        gpu.set_option("pml_variant", -1)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=4)
    assert (out == exp).all()
    assert (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0)


@pytest.mark.parametrize("blocks", [1, 3, 7])
def test_ragged_batches_vs_oracle(engines, blocks):
    """Batches built to stress lanes that start and finish out of step (written for the lane-refill kernel, which round 5
    removed; kept for the default walk): empty reads between long ones, a run of empty reads longer than a wavefront, one-base
    reads, fewer reads than lanes in the last wavefront, illegal bases."""
    from oracle import build_index as B
    gpu, cpu = engines[6]
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(1300 + blocks)
    reads = mutated_reads(rng, ref, 500, 1, 40) + [b""] * 150 + mutated_reads(rng, ref, 300, 150, 400)
    reads += [b"A", b"", b"N", b"", b"ACGTN" * 7] * 20 + mutated_reads(rng, ref, 30, 1500, 2500)
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order[:900]] + [b""] * 70 + [reads[i] for i in order[900:]]
    bases, offs = pack(reads)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=4)
    gpu.set_option("waves_per_cu", blocks)        # (also: the occupancy cap at 1 / 3 / 7 wavefronts per CU)
    try:
        out, st, err, rc = gpu.query_pml_packed(bases, offs, want_err=True)
    finally:
        gpu.set_option("waves_per_cu", 0)
    assert rc == 0 and not err.any()
    assert (out == exp).all()
    assert (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0)


@pytest.mark.parametrize("mode", [6, 8])
def test_count_vs_oracle(engines, mode):
    from oracle import build_index as B
    gpu, cpu = engines[mode]
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(200 + mode)
    reads = mutated_reads(rng, ref, 500, 1, 400)
    reads += [b"", b"A", b"N", b"AN", b"NA", b"ACGTN", b"C" * 40, ref[1000:1300], ref[5:6]]
    bases, offs = pack(reads)
    m, c, st = gpu.query_count_packed(bases, offs)
    em, ec = cpu.count_batch(bases, offs, threads=4)
    assert (m == em).all() and (c == ec).all()
    assert st.errors == 0


def golden_image_bytes(mode):
    name = {6: "index_regular-thresholds", 8: "index_blocked-thresholds"}[mode]
    with open(os.path.join(GOLDEN, name, "index.movi"), "rb") as f:
        return f.read()


@pytest.mark.parametrize("mode", [6, 8])
def test_count_state_machine_vs_oracle(engines, mode):
    """"count_variant" 1 (round 5): the backward search as a lane state machine over row windows (zml_kernel_flat<..., CNT = 1>;
    by itself on tables beyond the TLBs' reach) against the oracle and against count_kernel_v0 -- matched lengths, counts, error
    bytes and the fast-forward / scan counters -- with and without the interval table (every K the table can have: the search
    then starts at base K), the pair-shared gathers, the 64-bit row indexes; reads of every length from 0, illegal bases at every
    distance from the read's end, bases that do not extend, exact substrings."""
    from oracle import build_index as B
    gpu, cpu = engines[mode]
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(2200 + mode)
    reads = mutated_reads(rng, ref, 700, 1, 400)
    reads += [b"", b"A", b"N", b"AN", b"NA", b"ACGTN", b"NACGT", b"C" * 40, b"T" * 300, ref[1000:1300], ref[5:6], ref[:700], ref[-700:]]
    reads += [ref[3000:3000 + L] for L in range(0, 41)]
    for pos in range(0, 20):                                   # an illegal base at distance pos from the read's end
        r = bytearray(ref[5000:5060]); r[59 - pos] = ord("N"); reads.append(bytes(r))
        r = bytearray(ref[5000:5060]); r[59 - pos] = ord("a"); reads.append(bytes(r))
    bases, offs = pack(reads)
    em, ec = cpu.count_batch(bases, offs, threads=4)
    try:
        for K in (12, 0, 1, 5, 7):
            gpu.set_option("ftab_k", K)
            gpu.set_option("count_variant", 0)
            m0, c0, st0 = gpu.query_count_packed(bases, offs)
            assert gpu.last_launch()["kernel"].startswith("count_kernel_v0<")
            assert (m0 == em).all() and (c0 == ec).all() and st0.errors == 0, K
            gpu.set_option("count_variant", 1)
            for idx64, pair in ((0, 0), (0, 1), (1, 0), (1, 1)):
                gpu.set_option("idx64", idx64)
                gpu.set_option("pair_loads", pair)
                m, c, st, err, rc = gpu.query_count_packed(bases, offs, want_err=True)
                li = gpu.last_launch()
                assert li["kernel"] == "zml_kernel_flat<6, %s, 0, 0, %d, 1>" % ("unsigned long" if idx64 else "unsigned int", pair), li
                assert rc == 0 and not err.any()
                assert (m == em).all() and (c == ec).all(), (K, idx64, pair)
                assert (st.fast_forwards, st.scans, st.errors) == (st0.fast_forwards, st0.scans, 0), (K, idx64, pair)
            gpu.set_option("idx64", 0)
    finally:
        gpu.set_option("idx64", 0)
        gpu.set_option("pair_loads", -1)
        gpu.set_option("count_variant", -1)
        gpu.set_option("ftab_k", 12)
    if mode == 6:
        # a separators index, and rows that point past the table: the search runs into the reference's throw (move_structure.cpp:63-65)
        # exactly where count_kernel_v0 does -- error bytes, matched lengths, counts and the counters' error tally
        import movi_amd
        from oracle.oracle import Oracle
        simg = B.build_index_from_seqs([ref[:40000], ref[40000:90000], ref[90000:]], 6, separators=True)
        gs, cs = movi_amd.MoveIndex.from_image(simg), Oracle(simg)
        sreads = reads[:800] + [bytes(ref[39950:40050]), b"ACG%TACGTACGTACGT", bytes(ref[100:1100])]
        sb, so = pack(sreads)
        sem, sec = cs.count_batch(sb, so, threads=4)
        for cv, pair in ((0, 0), (1, 0), (1, 1)):
            gs.set_option("count_variant", cv)
            gs.set_option("pair_loads", pair)
            m, c, st = gs.query_count_packed(sb, so)
            assert (m == sem).all() and (c == sec).all() and st.errors == 0, (cv, pair)
        gs.close()
        cs.close()
        bad = bytearray(golden_image_bytes(6))
        _, _, off, _ = movi_amd.parse_index_image(bytes(bad))
        rows = np.frombuffer(bad, np.uint8, count=118209 * 8, offset=off).reshape(-1, 8).copy()
        rows[np.random.default_rng(77).choice(118209, 3000, replace=False), 0:4] = 0xFF
        bad[off: off + rows.size] = rows.tobytes()
        gb = movi_amd.MoveIndex.from_image(bytes(bad))
        gb.set_option("count_variant", 0)
        m0, c0, st0, err0, rc0 = gb.query_count_packed(bases, offs, want_err=True)
        assert st0.errors > 50
        for pair in (0, 1):
            gb.set_option("count_variant", 1)
            gb.set_option("pair_loads", pair)
            m, c, st, err, rc = gb.query_count_packed(bases, offs, want_err=True)
            assert gb.last_launch()["kernel"] == "zml_kernel_flat<6, unsigned int, 0, 0, %d, 1>" % pair
            assert rc == rc0 and (err == err0).all() and (m == m0).all() and (c == c0).all() and st.errors == st0.errors, pair
        gb.close()
    if mode == 6:                                              # the threshold-less layouts (`regular`, `blocked` -> kmode 3) run it too
        import movi_amd
        from oracle.oracle import Oracle
        for tmode in (3, 2):
            img = B.build_index_from_seqs([ref], tmode)
            g3, c3 = movi_amd.MoveIndex.from_image(img), Oracle(img)
            em3, ec3 = c3.count_batch(bases, offs, threads=4)
            g3.set_option("count_variant", 1)
            for pair in (0, 1):
                g3.set_option("pair_loads", pair)
                m, c, st = g3.query_count_packed(bases, offs)
                assert g3.last_launch()["kernel"] == "zml_kernel_flat<3, unsigned int, 0, 0, %d, 1>" % pair
                assert (m == em3).all() and (c == ec3).all() and st.errors == 0, (tmode, pair)
            g3.close()
            c3.close()


@pytest.mark.parametrize("mode", [6, 8])
@pytest.mark.parametrize("variant", [0, 1])
def test_zml_vs_oracle(engines, mode, variant):
    """MoveStructure::query_zml (src/move_structure_query.cpp:690-785): ragged reads with substitutions
    and illegal characters, the edge cases around illegal first / last bases, every length 0..40
    (packed-store tails)."""
    from oracle import build_index as B
    gpu, cpu = engines[mode]
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(300 + mode)
    reads = mutated_reads(rng, ref, 600, 1, 500)
    reads += [b"", b"A", b"N", b"NN", b"AN", b"NA", b"NAN", b"ACGTN", b"NACGT", b"ACGTNNACGT", b"aACGT", b"ACGTa",
              b"C" * 40, ref[1000:1300], ref[5:6], ref[2000:2300] + b"N" + ref[7000:7100]]
    reads += [ref[3000:3000 + L] for L in range(0, 41)]
    reads += [s for _, s in read_fastx(os.path.join(GOLDEN, "sample.fastq"))]
    bases, offs = pack(reads)
    gpu.set_option("zml_variant", variant)                   # 0 = base-synchronous kernel, 1 = lane state machine
    try:
        z, st = gpu.query_zml_packed(bases, offs)
        assert (z == cpu.zml_batch(bases, offs, threads=4)).all()
        assert st.errors == 0
        for i in (0, 5, 600, 601, 602, 603, 604, 605, 606):
            assert (gpu.query_zml([reads[i]])[0] == cpu.zml(reads[i])).all()
        # an exact substring is one phrase: 0, 1, 2, ...
        one = gpu.query_zml([ref[1000:1300]])[0]
        assert (one == np.arange(300)).all()
        # the two kernels walk the same rows: identical fast-forward and scan counts
        gpu.set_option("zml_variant", 1 - variant)
        z2, st2 = gpu.query_zml_packed(bases, offs)
        assert (z2 == z).all() and (st2.fast_forwards, st2.scans) == (st.fast_forwards, st.scans)
    finally:
        gpu.set_option("zml_variant", -1)


def test_zml_u16_clamp_on_device(built_lib):
    import movi_amd
    from oracle import build_index as B
    img = B.build_index_from_seqs([b"A" * 70000], 6, rc=False)
    gpu = movi_amd.MoveIndex.from_image(img)
    z = gpu.query_zml([b"A" * 66000])[0]
    assert z[0] == 0 and z[65535] == 65535 and z[-1] == 65535
    assert (np.diff(z[:65536].astype(np.int64)) == 1).all()


@pytest.mark.parametrize("mode", [6, 8])
def test_large_zml_batch_properties(built_lib, mode):
    """1 M x 150 bp --zml on a 10 M-row table: size-independent properties on everything (values
    restart at 0 and grow by exactly 1 inside a phrase; exact walks are a single phrase), the oracle
    on a slice."""
    import movi_amd
    from oracle.oracle import Oracle
    from tools import synth
    six = synth.synth_index(10_000_000, mode=mode, seed=78)
    img = six.image()
    gpu = movi_amd.MoveIndex.from_image(img)
    n_reads, L = 1_000_000, 150
    bases, offs = synth.synth_reads(six, n_reads, L, seed=15, sub_rate=0.01, n_rate=0.001)
    z, st = gpu.query_zml_packed(bases, offs)
    assert st.errors == 0
    zz = z.reshape(n_reads, L).astype(np.int32)
    assert (zz[:, 0] == 0).all()
    d = np.diff(zz, axis=1)
    assert ((d == 1) | (zz[:, 1:] == 0)).all()          # either the phrase grows by one or a new one starts
    illegal = bases.reshape(n_reads, L)[:, ::-1] == ord("N")
    assert (zz[illegal] == 0).all()
    cpu = Oracle(img)
    sl = slice(250_000, 253_000)
    exp = cpu.zml_batch(bases[sl.start * L: sl.stop * L], offs[sl.start: sl.stop + 1] - offs[sl.start], threads=8)
    assert (z[sl.start * L: sl.stop * L] == exp).all()


@pytest.mark.parametrize("mode", [6, 8])
def test_synthetic_index_vs_oracle(built_lib, mode):
    """Seeded synthetic table (tools/synth.py) at a size the oracle does in seconds."""
    import movi_amd
    from oracle.oracle import Oracle
    from tools import synth
    six = synth.synth_index(300000, mode=mode, seed=42 + mode)
    img = six.image()
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    lens = np.random.default_rng(3).integers(1, 600, size=5000)
    bases, offs = synth.synth_reads(six, 5000, 0, seed=9, lens=lens)
    out, st = gpu.query_pml_packed(bases, offs)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=8)
    assert (out == exp).all() and (st.fast_forwards, st.scans) == (ff, sc)
    m, c, _ = gpu.query_count_packed(bases, offs)
    em, ec = cpu.count_batch(bases, offs, threads=8)
    assert (m == em).all() and (c == ec).all()
    z, zst = gpu.query_zml_packed(bases, offs)
    assert (z == cpu.zml_batch(bases, offs, threads=8)).all() and zst.errors == 0


def test_u16_clamp_on_device(built_lib):
    """MoveQuery::add_ml clamp (include/move_query.hpp:26-38) on a homopolymer index."""
    import movi_amd
    from oracle import build_index as B
    img = B.build_index_from_seqs([b"A" * 70000], 6, rc=False)
    gpu = movi_amd.MoveIndex.from_image(img)
    p = gpu.query_pml([b"A" * 66000])[0]
    assert p[0] == 1 and p[65534] == 65535 and p[-1] == 65535


def test_host_argument_validation(engines):
    """Bad offsets are refused with MOVI_ERR_ARG (-1), not walked."""
    from movi_amd._lib import MoviError
    gpu, _ = engines[6]
    bases = np.frombuffer(b"ACGTACGTAC", np.uint8)
    for q in (gpu.query_pml_packed, gpu.query_zml_packed, gpu.query_count_packed):
        with pytest.raises(MoviError) as e:
            q(bases, np.array([0, 6, 4, 10], np.uint64))
        assert e.value.code == -1 and "non-decreasing" in str(e.value)
        # a decreasing pair that would underflow the chunk's byte count: refused before anything is allocated
        with pytest.raises(MoviError) as e:
            q(bases, np.array([100, 0], np.uint64))
        assert e.value.code == -1 and "non-decreasing" in str(e.value)
    with pytest.raises(MoviError) as e:
        gpu.classify_packed(bases, np.array([100, 0], np.uint64), 150, 8)
    assert e.value.code == -1


def test_launch_options_are_bounded(engines):
    """block_threads beyond the kernels' __launch_bounds__(256) is refused; the occupancy cap (dynamic LDS padding
    above 64 KiB needs an opt-in per kernel) works with every selectable PML kernel."""
    from movi_amd._lib import MoviError
    from oracle import build_index as B
    gpu, cpu = engines[6]
    for bad in (-64, 32, 100, 512, 1024):
        with pytest.raises(MoviError) as e:
            gpu.set_option("block_threads", bad)
        assert e.value.code == -1
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    reads = mutated_reads(np.random.default_rng(77), ref, 400, 1, 200)
    bases, offs = pack(reads)
    exp, _, _ = cpu.pml_batch(bases, offs, threads=4)
    try:
        for bt in (64, 128, 256):
            gpu.set_option("block_threads", bt)
            for wpc in (1, 2, 4):                             # 159 KiB / 79 KiB / 39 KiB of dynamic LDS per block of 64
                gpu.set_option("waves_per_cu", wpc)
                for variant in (1, 14):
                    gpu.set_option("pml_variant", variant)
                    out, st = gpu.query_pml_packed(bases, offs)
                    assert (out == exp).all() and st.errors == 0, (bt, wpc, variant)
    finally:
        gpu.set_option("pml_variant", -1)
        gpu.set_option("waves_per_cu", 0)
        gpu.set_option("block_threads", 0)


def test_invariant_violation_is_flagged_not_hidden(built_lib, golden_image):
    """A corrupted table must surface as MOVI_ERR_INVARIANT + per-read flags (the
    reference throws, src/move_structure.cpp:63-65)."""
    import movi_amd
    img = bytearray(golden_image(6))
    _, _, off, _ = movi_amd.parse_index_image(bytes(img))
    rows = np.frombuffer(img, np.uint8, count=118209 * 8, offset=off).reshape(-1, 8)
    rows = rows.copy()
    rows[:, 0:4] = 0xFF                                   # every destination id >= r
    img[off: off + rows.size] = rows.tobytes()
    gpu = movi_amd.MoveIndex.from_image(bytes(img))
    bases, offs = pack([b"ACGTACGT", b"A"])
    out, st, err, rc = gpu.query_pml_packed(bases, offs, want_err=True)
    assert rc == -6 and st.errors == 1 and list(err) == [1, 0]


def test_large_batch_properties(built_lib):
    """BASELINE-sized shape (1 M x 150 bp on a 10 M-row table) checked through
    size-independent properties plus an oracle spot check of a 2 k-read slice."""
    import movi_amd
    from oracle.oracle import Oracle
    from tools import synth
    six = synth.synth_index(10_000_000, mode=6, seed=20260529)
    img = six.image()
    gpu = movi_amd.MoveIndex.from_image(img)
    n_reads, L = 1_000_000, 150
    bases, offs = synth.synth_reads(six, n_reads, L, seed=1)
    out, st = gpu.query_pml_packed(bases, offs)
    assert st.errors == 0 and st.bases == n_reads * L
    pm = out.reshape(n_reads, L).astype(np.int64)
    rb = bases.reshape(n_reads, L)[:, ::-1]                # emission order
    # (1) an illegal base has PML 0; (2) PML either resets to 0/1.. or grows by exactly 1
    assert (pm[rb == ord("N")] == 0).all()
    d = pm[:, 1:] - pm[:, :-1]
    assert ((d == 1) | (pm[:, 1:] == 0)).all()
    # (3) idempotence / determinism: a second pass gives identical bytes
    out2, _ = gpu.query_pml_packed(bases, offs)
    assert (out == out2).all()
    # (4) batch-composition independence: any slice alone gives the same PMLs
    sl = slice(123_000, 125_000)
    cpu = Oracle(img)
    sb = bases[int(offs[sl.start]): int(offs[sl.stop])]
    so = offs[sl.start: sl.stop + 1] - offs[sl.start]
    exp, _, _ = cpu.pml_batch(sb, so, threads=8)
    assert (out[int(offs[sl.start]): int(offs[sl.stop])] == exp).all()
    sub, _ = gpu.query_pml_packed(sb, so)
    assert (sub == exp).all()


def test_host_chunk_boundary(built_lib):
    """The host entry points cut their input into launches of >= 2^28 bases AND >= 2^18 reads (long reads: up to
    2^31 bases).  300 k x 1 kbp crosses one cut (after read ~268 k): the result must not depend on it."""
    import movi_amd
    from oracle.oracle import Oracle
    from tools import synth
    six = synth.synth_index(1_000_000, mode=6, seed=4)
    img = six.image()
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    n_reads, L = 300_000, 1000
    bases, offs = synth.synth_reads(six, n_reads, L, seed=8, sub_rate=0.03)
    out, st = gpu.query_pml_packed(bases, offs)
    assert st.errors == 0 and st.bases == n_reads * L
    for lo in (0, 262_000, 268_000, 299_000):               # around 2^18 reads and 2^28 bases
        hi = lo + 1000
        exp, _, _ = cpu.pml_batch(bases[lo * L: hi * L], offs[lo: hi + 1] - offs[lo], threads=8)
        assert (out[lo * L: hi * L] == exp).all()
    z, _ = gpu.query_zml_packed(bases, offs)
    lo, hi = 267_500, 269_000
    assert (z[lo * L: hi * L] == cpu.zml_batch(bases[lo * L: hi * L], offs[lo: hi + 1] - offs[lo], threads=8)).all()
    m, c, _ = gpu.query_count_packed(bases, offs)
    em, ec = cpu.count_batch(bases[lo * L: hi * L], offs[lo: hi + 1] - offs[lo], threads=8)
    assert (m[lo:hi] == em).all() and (c[lo:hi] == ec).all()


def test_long_read_batch_properties_multi_chunk(built_lib):
    """BASELINE config 3 shape (100 k x 10 kbp = 1 Gbase on a 10 M-row table): one launch of the host
    path (long reads extend a chunk until it holds 2^18 reads or 2^31 bases), the low-occupancy
    kernel; checked by properties + an oracle spot check."""
    import movi_amd
    from oracle.oracle import Oracle
    from tools import synth
    six = synth.synth_index(10_000_000, mode=6, seed=20260529)
    img = six.image()
    gpu = movi_amd.MoveIndex.from_image(img)
    n_reads, L = 100_000, 10_000
    bases, offs = synth.synth_reads(six, n_reads, L, seed=3, sub_rate=0.08)
    out, st = gpu.query_pml_packed(bases, offs)
    assert st.errors == 0 and st.bases == n_reads * L
    pm = out.reshape(n_reads, L)
    rb = bases.reshape(n_reads, L)[:, ::-1]
    assert (pm[rb == ord("N")] == 0).all()
    d = pm[:, 1:].astype(np.int32) - pm[:, :-1].astype(np.int32)
    assert ((d == 1) | (pm[:, 1:] == 0)).all()
    cpu = Oracle(img)
    for i in (0, 26843, 26844, 53687, 99_999):                 # reads next to the chunk boundaries
        assert (pm[i] == cpu.pml(bases[i * L:(i + 1) * L].tobytes())).all()


@pytest.mark.parametrize("mode", [6, 8])
def test_large_count_batch_properties(built_lib, mode):
    """BASELINE config 5 shape at single-GPU scale (1 M x 150 bp --count): exact substrings are
    found end to end; mutated reads agree with the oracle on a slice."""
    import movi_amd
    from oracle.oracle import Oracle
    from tools import synth
    six = synth.synth_index(10_000_000, mode=mode, seed=77)
    img = six.image()
    gpu = movi_amd.MoveIndex.from_image(img)
    n_reads, L = 1_000_000, 150
    bases, offs = synth.synth_reads(six, n_reads, L, seed=5, sub_rate=0.0, n_rate=0.0)
    m, c, st = gpu.query_count_packed(bases, offs)
    assert st.errors == 0
    cpu = Oracle(img)
    short = np.flatnonzero(m != L)          # a walk through the terminator row is not a substring of the text
    assert short.size <= 5
    for i in short:
        assert cpu.count(bases[i * L:(i + 1) * L].tobytes()) == (int(m[i]), int(c[i]))
    assert (c >= 1).all()
    bases2, offs2 = synth.synth_reads(six, n_reads, L, seed=6, sub_rate=0.02, n_rate=0.001)
    m2, c2, _ = gpu.query_count_packed(bases2, offs2)
    assert (m2 <= L).all() and ((m2 > 0) | (bases2.reshape(n_reads, L)[:, -1] == ord("N"))).all()
    sl = slice(500_000, 503_000)
    em, ec = cpu.count_batch(bases2[sl.start * L: sl.stop * L], offs2[sl.start: sl.stop + 1] - offs2[sl.start], threads=8)
    assert (m2[sl] == em).all() and (c2[sl] == ec).all()


@pytest.mark.parametrize("bin_width,thr", [(150, 7), (40, 4), (1, 1), (1000, 20)])
def test_classification_bins_on_device(engines, bin_width, thr):
    """movi_pml_classify_host == Classifier::classify's bins (src/classifier.cpp:99-143) applied to the
    oracle's PML vectors: bins in emission order, last bin absorbs a short remainder."""
    from oracle import build_index as B
    gpu, cpu = engines[6]
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(bin_width + thr)
    reads = mutated_reads(rng, ref, 400, 1, 1200) + [b"", b"A", ref[:149], ref[:150], ref[:151], ref[:299], ref[:300], ref[:449]]
    bases, offs = pack(reads)
    a, b, s = gpu.classify_packed(bases, offs, bin_width, thr)
    for i, r in enumerate(reads):
        p = cpu.pml(r)
        n, start, ea, eb, es = len(p), 0, 0, 0, 0
        while start < n:
            end = start + bin_width if start + bin_width < n else n
            if n - end < bin_width:
                end = n
            mx = int(p[start:end].max())
            ea += mx >= thr
            eb += mx < thr
            es += mx
            start = end
        assert (int(a[i]), int(b[i]), int(s[i])) == (ea, eb, es), (i, len(r))


@pytest.mark.parametrize("mode", [6, 8])
@pytest.mark.parametrize("variant", [1, 14])
def test_fused_classification_kernels(engines, mode, variant):
    """movi_pml_classify_device: the bins fused into the PML walk, with and without the PML vector, in both
    shipped kernels, against the bins of the oracle's PML vectors and against the standalone
    movi_classify_device reduction over the resident vectors."""
    import ctypes as C
    import torch
    from conftest import classify_py
    from movi_amd._lib import lib
    from oracle import build_index as B
    gpu, cpu = engines[mode]
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(900 + mode + variant)
    reads = mutated_reads(rng, ref, 700, 1, 900) + [b"", b"A", ref[:149], ref[:150], ref[:151], ref[:299], ref[:300], ref[:449]]
    bases, offs = pack(reads)
    n = len(reads)
    dev = torch.device("cuda", 0)
    d_bases = torch.from_numpy(bases.copy()).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64).copy()).to(dev)
    exp_pml, _, _ = cpu.pml_batch(bases, offs, threads=4)
    gpu.set_option("pml_variant", variant)
    try:
        for bin_width, thr in ((150, 7), (40, 3), (1, 1)):
            exp = [classify_py(exp_pml[int(offs[i]):int(offs[i + 1])], thr, bin_width) if len(r) else None
                   for i, r in enumerate(reads)]
            for with_vector in (True, False):
                d_out = torch.zeros(max(bases.size, 1), dtype=torch.int16, device=dev)
                d_a = torch.full((n,), -1, dtype=torch.int32, device=dev)
                d_b = torch.full((n,), -1, dtype=torch.int32, device=dev)
                d_s = torch.full((n,), -1, dtype=torch.int64, device=dev)
                gpu.pml_classify_device(d_bases.data_ptr(), d_offs.data_ptr(), n, bases.size, bin_width, thr,
                                        d_out.data_ptr() if with_vector else 0, d_a.data_ptr(), d_b.data_ptr(),
                                        d_s.data_ptr())
                torch.cuda.synchronize()
                a, b, sm = d_a.cpu().numpy(), d_b.cpu().numpy(), d_s.cpu().numpy()
                for i, e in enumerate(exp):
                    if e is None:
                        assert (a[i], b[i], sm[i]) == (0, 0, 0)
                    else:
                        found, avg, ea, eb = e
                        assert (a[i], b[i]) == (ea, eb) and sm[i] == round(avg * (ea + eb)), (i, len(reads[i]))
                if with_vector:
                    assert (d_out.cpu().numpy().view(np.uint16)[:bases.size] == exp_pml).all()
                    # standalone reduction over the resident vectors agrees
                    d_a2, d_b2, d_s2 = torch.zeros_like(d_a), torch.zeros_like(d_b), torch.zeros_like(d_s)
                    rc = lib().movi_classify_device(gpu._h, C.c_void_p(d_out.data_ptr()), C.c_void_p(d_offs.data_ptr()), n,
                                                    bin_width, thr, C.c_void_p(d_a2.data_ptr()), C.c_void_p(d_b2.data_ptr()),
                                                    C.c_void_p(d_s2.data_ptr()), None)
                    assert rc == 0
                    torch.cuda.synchronize()
                    nz = np.array([len(r) > 0 for r in reads])
                    assert (d_a2.cpu().numpy()[nz] == a[nz]).all() and (d_b2.cpu().numpy()[nz] == b[nz]).all()
                    assert (d_s2.cpu().numpy()[nz] == sm[nz]).all()
                else:
                    assert int(d_out.abs().sum().item()) == 0          # nothing was written
    finally:
        gpu.set_option("pml_variant", -1)


@pytest.mark.parametrize("alphabet", [b"ACGT", b"ACG", b"AT", b"GT", b"C"])
def test_fuzz_small_indexes(built_lib, tmp_path, alphabet):
    """Many tiny indexes with awkward structure (reduced alphabets -> shifted codes and a shorter
    base-interval table, long runs split at MAX_RUN_LENGTH, repeats, the terminator row in odd
    places), built by tools/build_index; PML + count on the GPU vs the oracle, both modes."""
    import subprocess
    import movi_amd
    from oracle.oracle import Oracle
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "build_index")
    if not os.path.exists(tool):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", tool, tool + ".cpp"])
    rng = np.random.default_rng(len(alphabet) * 1000 + alphabet[0])
    for trial in range(6):
        recs = []
        for _ in range(int(rng.integers(1, 4))):
            unit = bytes(rng.choice(list(alphabet), size=int(rng.integers(5, 400))).astype(np.uint8))
            s = unit * int(rng.integers(1, 6)) + bytes([alphabet[0]]) * int(rng.integers(0, 3000))
            recs.append(s)
        fa = tmp_path / ("f%d.fa" % trial)
        fa.write_bytes(b"".join(b">s%d\n%s\n" % (i, s) for i, s in enumerate(recs)))
        text = b"".join(recs)
        reads = []
        for _ in range(120):
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
            for variant in (1, 14):
                gpu.set_option("pml_variant", variant)
                out, st = gpu.query_pml_packed(bases, offs)
                exp, ff, sc = cpu.pml_batch(bases, offs, threads=2)
                assert (out == exp).all(), (alphabet, trial, mode, variant)
                assert (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0)
            check_segmented(gpu, bases, offs, exp, ff, sc, (alphabet, trial, mode))
            m, c, _ = gpu.query_count_packed(bases, offs)
            em, ec = cpu.count_batch(bases, offs, threads=2)
            assert (m == em).all() and (c == ec).all(), (alphabet, trial, mode)
            z, _ = gpu.query_zml_packed(bases, offs)
            assert (z == cpu.zml_batch(bases, offs, threads=2)).all(), (alphabet, trial, mode)
            gpu.close()


@pytest.mark.parametrize("mode", [6, 8, 7])
def test_real_bwt_pangenome_vs_oracle(built_lib, tmp_path, mode):
    """A real-BWT index of a synthetic 16-genome pangenome (~1 M rows, built by tools/build_index) with
    20 k substrings + mutations: every PML and every count against the oracle (which walks the index type as
    stored: blocked ids, sampled ids through get_id per step), on the GPU through the expanded table."""
    import subprocess
    import movi_amd
    from oracle.oracle import Oracle
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "build_index")
    if not os.path.exists(tool):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", tool, tool + ".cpp"])
    out = str(tmp_path / "pg")
    subprocess.check_call([tool, "pangenome", "500000", "16", "0.002", "5", str(mode), out, "20000", "150", "0.02"],
                          stderr=subprocess.DEVNULL)
    img = open(os.path.join(out, "index.movi"), "rb").read()
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    assert gpu.desc.r > 500_000
    bases = np.fromfile(os.path.join(out, "reads.bin"), np.uint8)
    offs = (np.arange(20001, dtype=np.uint64) * np.uint64(150))
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=8)
    for variant in (1, 14):
        gpu.set_option("pml_variant", variant)
        got, st = gpu.query_pml_packed(bases, offs)
        assert (got == exp).all() and (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0)
    m, c, _ = gpu.query_count_packed(bases, offs)
    em, ec = cpu.count_batch(bases, offs, threads=8)
    assert (m == em).all() and (c == ec).all()
    z, zst = gpu.query_zml_packed(bases, offs)
    assert (z == cpu.zml_batch(bases, offs, threads=8)).all() and zst.errors == 0


# ------------------------------------------------------------------ separators ('%' + ACGT) indexes
# movi build --separators: reference index-size KATs tests/test_build.cpp:79,95 (pinned on the CPU in
# tests/test_separators_cpu.py, which also holds the oracle to a BWT-level simulation).

@pytest.mark.parametrize("mode", [6, 8])
def test_separators_reference_index_vs_oracle(built_lib, mode):
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    img = B.build_index_from_seqs([ref], mode, separators=True)
    assert len(img) == {6: 948232, 8: 711854}[mode]
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    assert gpu.desc.alphabet_size == 5
    rng = np.random.default_rng(70 + mode)
    reads = mutated_reads(rng, ref, 400, 1, 1500)
    L = len(ref)
    clean = bytes(B.clean_text([ref], separators=True)[:L])
    rc = clean.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]
    reads += [clean[L - 40:] + rc[:40], clean[L - 5:] + b"%" + rc[:30], rc[-60:], b"%", b"A%C", b"%%ACGT", b"", b"N"]
    reads += [s for _, s in read_fastx(os.path.join(GOLDEN, "sample.fastq"))]
    bases, offs = pack(reads)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=4)
    for variant in (0, 1, 14):
        gpu.set_option("pml_variant", variant)
        out, st = gpu.query_pml_packed(bases, offs)
        assert (out == exp).all(), variant
        assert (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0)
    check_segmented(gpu, bases, offs, exp, ff, sc, "separators")
    m, c, _ = gpu.query_count_packed(bases, offs)
    em, ec = cpu.count_batch(bases, offs, threads=4)
    assert (m == em).all() and (c == ec).all()
    z, _ = gpu.query_zml_packed(bases, offs)
    assert (z == cpu.zml_batch(bases, offs, threads=4)).all()
    gpu.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_separators_fuzz_many_sequences(built_lib, seed):
    """Many short records: many rows of the separator (side-table thresholds), walks that reposition on them."""
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    from test_separators_cpu import _multi_sequence_case
    rng = np.random.default_rng(4100 + seed)
    seqs, t, reads = _multi_sequence_case(rng, int(rng.integers(2, 60)))
    reads += [b"", b"%", b"N%N"]
    bases, offs = pack(reads)
    bwt, thr = B.bwt_and_thresholds(t)
    for mode in (6, 8):
        img = B.serialize(B.build_rows(bwt, thr, mode))
        gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
        exp, ff, sc = cpu.pml_batch(bases, offs, threads=2)
        for variant in (0, 1, 14):
            gpu.set_option("pml_variant", variant)
            out, st = gpu.query_pml_packed(bases, offs)
            assert (out == exp).all(), (seed, mode, variant)
            assert (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0)
        check_segmented(gpu, bases, offs, exp, ff, sc, (seed, mode))
        m, c, _ = gpu.query_count_packed(bases, offs)
        em, ec = cpu.count_batch(bases, offs, threads=2)
        assert (m == em).all() and (c == ec).all(), (seed, mode)
        z, _ = gpu.query_zml_packed(bases, offs)
        assert (z == cpu.zml_batch(bases, offs, threads=2)).all(), (seed, mode)
        gpu.close()


# ------------------------------------------------------------------ sampled-thresholds (mode 7) indexes
# 3-byte rows without ids + checkpointed id table (reference KATs tests/test_build.cpp:45-47,86-88; the
# reference's PML golden test covers this index type too, tests/test_pml.cpp:98-100).

@pytest.mark.parametrize("separators", [False, True])
def test_sampled_thresholds_index_vs_oracle(built_lib, separators):
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    img = B.build_index_from_seqs([ref], 7, separators=separators)
    assert len(img) == (505009 if separators else 475326)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    assert gpu.desc.mode == 7 and gpu.desc.row_bytes == 3
    golden_reads = [s for _, s in read_fastx(os.path.join(GOLDEN, "sample.fastq"))]
    if not separators:
        gold_pml, _ = golden_sorted_pmls()
        assert sorted(stdout_line(p) for p in gpu.query_pml(golden_reads)) == gold_pml
    rng = np.random.default_rng(170 + separators)
    reads = mutated_reads(rng, ref, 500, 1, 1500) + golden_reads + [b"", b"A", b"N", b"%", b"ACGT" * 200]
    bases, offs = pack(reads)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=4)
    out, st = gpu.query_pml_packed(bases, offs)
    assert (out == exp).all()
    assert (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0)
    m, c, _ = gpu.query_count_packed(bases, offs)
    em, ec = cpu.count_batch(bases, offs, threads=4)
    assert (m == em).all() and (c == ec).all()
    z, _ = gpu.query_zml_packed(bases, offs)
    assert (z == cpu.zml_batch(bases, offs, threads=4)).all()
    gpu.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_sampled_thresholds_fuzz(built_lib, seed):
    """Small awkward texts (repeats, long runs split at 511, few rows per checkpoint span, the terminator row and
    the table end inside a span), with and without separators."""
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    rng = np.random.default_rng(7100 + seed)
    seqs = []
    for _ in range(int(rng.integers(1, 5))):
        unit = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(3, 300))).astype(np.uint8))
        seqs.append(unit * int(rng.integers(1, 8)) + b"A" * int(rng.integers(0, 2500)))
    text = b"".join(seqs)
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
    for separators in (False, True):
        img = B.build_index_from_seqs(seqs, 7, separators=separators)
        gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
        exp, ff, sc = cpu.pml_batch(bases, offs, threads=2)
        out, st = gpu.query_pml_packed(bases, offs)
        assert (out == exp).all(), (seed, separators)
        assert (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0)
        m, c, _ = gpu.query_count_packed(bases, offs)
        em, ec = cpu.count_batch(bases, offs, threads=2)
        assert (m == em).all() and (c == ec).all(), (seed, separators)
        z, _ = gpu.query_zml_packed(bases, offs)
        assert (z == cpu.zml_batch(bases, offs, threads=2)).all(), (seed, separators)
        gpu.close()


@pytest.mark.parametrize("mode", [6, 7, 8])
def test_64bit_index_instantiations(built_lib, golden_image, mode):
    """Tables of 2^32 rows and more run the uint64_t instantiations of the state-machine kernels and of the sampled
    mode's get_id; no test table is that large, so the "idx64" option routes a small index through them."""
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    img = golden_image(mode) if mode != 7 else B.build_index_from_seqs([ref], 7)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    rng = np.random.default_rng(640 + mode)
    reads = mutated_reads(rng, ref, 300, 1, 1200) + [b"", b"A", b"N", b"ACGT" * 100]
    bases, offs = pack(reads)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=4)
    gpu.set_option("idx64", 1)
    for variant in ((1, 14) if mode != 7 else (-1,)):
        gpu.set_option("pml_variant", variant)
        out, st = gpu.query_pml_packed(bases, offs)
        assert (out == exp).all(), (mode, variant)
        assert (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0)
    m, c, _ = gpu.query_count_packed(bases, offs)
    em, ec = cpu.count_batch(bases, offs, threads=4)
    assert (m == em).all() and (c == ec).all()
    gpu.set_option("idx64", 0)
    gpu.set_option("pml_variant", -1)
    out, _ = gpu.query_pml_packed(bases, offs)
    assert (out == exp).all()
    gpu.close()


def test_sampled_expansion_equals_regular_rows(built_lib, golden_image):
    """The sampled index is expanded on the GPU to regular-thresholds rows by the reference's get_id.  On ref.fasta the
    511-base run cap of mode 7 splits nothing further (r = 118209 in both modes), so the expanded table must equal the row
    table of the KAT-pinned regular-thresholds index byte for byte: every id, offset, length, character and threshold bit."""
    import ctypes as C
    import movi_amd
    from oracle import build_index as B
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    gpu = movi_amd.MoveIndex.from_image(B.build_index_from_seqs([ref], 7))
    ptr, n = gpu.device_rows()
    assert n == 118209 * 8
    host = np.empty(n, np.uint8)
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(C.c_void_p(host.ctypes.data), C.c_void_p(ptr), C.c_size_t(n), C.c_int(2)) == 0   # device -> host
    img6 = np.frombuffer(golden_image(6), np.uint8)
    _, _, off, nbytes = movi_amd.parse_index_image(img6)
    assert nbytes == n
    assert (host == img6[off: off + nbytes]).all()
    gpu.close()


@pytest.mark.parametrize("separators", [False, True])
def test_sampled_no_thresholds_index(built_lib, separators):
    """Mode 5: count and ZML against the oracle; PML refused (the reference repositions randomly without thresholds); the
    table expanded on the GPU equals the same rows encoded in the regular-thresholds layout by the numpy constructor."""
    import ctypes as C
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    t = B.clean_text([ref], separators=separators)
    f = B.build_rows(*B.bwt_and_thresholds(t), 5)
    img = B.serialize(f)
    assert len(img) == (464203 if separators else 437006)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    assert gpu.desc.mode == 5
    rng = np.random.default_rng(500 + separators)
    reads = mutated_reads(rng, ref, 400, 1, 1200) + [b"", b"A", b"N", b"%", b"ACGT" * 100]
    bases, offs = pack(reads)
    m, c, _ = gpu.query_count_packed(bases, offs)
    em, ec = cpu.count_batch(bases, offs, threads=4)
    assert (m == em).all() and (c == ec).all()
    z, _ = gpu.query_zml_packed(bases, offs)
    assert (z == cpu.zml_batch(bases, offs, threads=4)).all()
    with pytest.raises(movi_amd.MoviError) as e:
        gpu.query_pml_packed(bases, offs)
    assert e.value.code == -1 and "thresholds" in str(e.value)
    ptr, n = gpu.device_rows()
    host = np.empty(n, np.uint8)
    assert C.CDLL("libamdhip64.so").hipMemcpy(C.c_void_p(host.ctypes.data), C.c_void_p(ptr), C.c_size_t(n), C.c_int(2)) == 0
    assert host.tobytes() == B.encode_rows(dict(f, mode=6))
    gpu.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_sampled_expansion_equals_oracle_get_id_fuzz(built_lib, seed):
    """get_id on the GPU for every row of awkward multi-record texts (modes 7 and 5, with and without separators): the
    expanded table holds the oracle's get_id of every row (the reference's algorithm, its blind spot when r is a multiple
    of the checkpoint distance included -- seed 2 has r = 600) next to the constructor's offset / length / character /
    threshold bits; wherever the reference's get_id works it is the constructor's own id."""
    import ctypes as C
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    rng = np.random.default_rng(8100 + seed)
    seqs = []
    for _ in range(int(rng.integers(1, 6))):
        unit = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(3, 200))).astype(np.uint8))
        seqs.append(unit * int(rng.integers(1, 8)) + b"T" * int(rng.integers(0, 3000)))
    hip = C.CDLL("libamdhip64.so")
    for separators in (False, True):
        bwt, thr = B.bwt_and_thresholds(B.clean_text(seqs, separators=separators))
        for mode in (7, 5):
            f = B.build_rows(bwt, thr, mode)
            img = B.serialize(f)
            gpu = movi_amd.MoveIndex.from_image(img)
            ptr, n = gpu.device_rows()
            host = np.empty(n, np.uint8)
            assert hip.hipMemcpy(C.c_void_p(host.ctypes.data), C.c_void_p(ptr), C.c_size_t(n), C.c_int(2)) == 0
            ids = Oracle(img).get_ids().astype(np.int64)
            if f["r"] % 20:                                    # the reference's get_id is sound: it is the constructor's id
                assert (ids == f["pp_id"]).all(), (seed, separators, mode)
            assert host.tobytes() == B.encode_rows(dict(f, mode=6, pp_id=ids)), (seed, separators, mode)
            gpu.close()


@pytest.mark.parametrize("mode", [3, 2])
@pytest.mark.parametrize("separators", [False, True])
def test_no_threshold_regular_and_blocked_indexes(built_lib, mode, separators):
    """`regular` (mode 3) and `blocked` (mode 2) indexes -- the reference's threshold-less 8- and 6-byte row types, KAT sizes
    871479 / 654253 B (tests/test_build.cpp:33,49): count and ZML against the oracle (itself pinned to those KATs and to the
    brute-force search); PML refused; the resident table equals the rows in the `regular` layout (blocked: every id
    reconstructed by get_id on the GPU)."""
    import ctypes as C
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    t = B.clean_text([ref], separators=separators)
    f = B.build_rows(*B.bwt_and_thresholds(t), mode)
    img = B.serialize(f)
    assert len(img) == {(3, False): 871479, (3, True): 871496, (2, False): 654253, (2, True): 654280}[(mode, separators)]
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    assert gpu.desc.mode == mode
    rng = np.random.default_rng(700 + mode + separators)
    reads = mutated_reads(rng, ref, 500, 1, 1200) + [b"", b"A", b"N", b"%", b"ACGT" * 100]
    bases, offs = pack(reads)
    m, c, _ = gpu.query_count_packed(bases, offs)
    em, ec = cpu.count_batch(bases, offs, threads=4)
    assert (m == em).all() and (c == ec).all()
    z, _ = gpu.query_zml_packed(bases, offs)
    assert (z == cpu.zml_batch(bases, offs, threads=4)).all()
    with pytest.raises(movi_amd.MoviError) as e:
        gpu.query_pml_packed(bases, offs)
    assert e.value.code == -1 and "thresholds" in str(e.value)
    with pytest.raises(movi_amd.MoviError):
        gpu.classify_packed(bases, offs, 150, 8)
    ptr, n = gpu.device_rows()
    assert n == f["r"] * 8
    host = np.empty(n, np.uint8)
    assert C.CDLL("libamdhip64.so").hipMemcpy(C.c_void_p(host.ctypes.data), C.c_void_p(ptr), C.c_size_t(n), C.c_int(2)) == 0
    assert host.tobytes() == B.encode_rows(dict(f, mode=3))
    gpu.set_option("idx64", 1)                               # the 64-bit-index instantiations of the same kernels
    m2, c2, _ = gpu.query_count_packed(bases, offs)
    assert (m2 == em).all() and (c2 == ec).all()
    gpu.close()


def test_blocked_index_with_many_blocks(built_lib, tmp_path):
    """A `blocked` table whose rows span several id blocks (block size forced down by shrinking BLOCK_SIZE in the numpy
    constructor): the check-point lookup of get_id is exercised for every (character, block) pair."""
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    rng = np.random.default_rng(4242)
    anc = rng.choice(np.frombuffer(b"ACGT", np.uint8), 60000).tobytes()
    old = dict(B.BLOCK_SIZE)
    try:
        B.BLOCK_SIZE[2] = 1 << 12
        B.BLOCK_SIZE[8] = 1 << 12
        for mode in (2, 8):
            img = B.build_index_from_seqs([anc], mode)
            gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
            assert gpu.desc.n_blocks > 4 and gpu.desc.block_size == 1 << 12
            reads = mutated_reads(rng, anc, 300, 1, 400)
            bases, offs = pack(reads)
            m, c, _ = gpu.query_count_packed(bases, offs)
            em, ec = cpu.count_batch(bases, offs, threads=4)
            assert (m == em).all() and (c == ec).all()
            z, _ = gpu.query_zml_packed(bases, offs)
            assert (z == cpu.zml_batch(bases, offs, threads=4)).all()
            gpu.close()
    finally:
        B.BLOCK_SIZE.update(old)


# ---------------------------------------------------------------- overlapped host path (page-locked caller buffers)

def _pinned_copy(a):
    import movi_amd
    p = movi_amd.pinned_empty(a.size, a.dtype)
    p[:] = a
    return p


@pytest.mark.parametrize("mode", [6, 8])
@pytest.mark.parametrize("chunk_bases", [0, 1, 3000, 100_000])
def test_overlapped_host_path_equals_synchronous(engines, mode, chunk_bases):
    """movi_*_host with the reads (and the PML / ZML vector) in page-locked memory run as chunks in flight on three
    streams; answers, error bytes and counters must be those of the synchronous path and of the oracle, whatever the cut
    (chunk_bases 1 = one read per chunk: every slot is reused many times)."""
    import movi_amd
    from oracle import build_index as B
    gpu, cpu = engines[mode]
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(4200 + mode)
    reads = mutated_reads(rng, ref, 300, 1, 1500) + [b"", b"", b"N", b"ACGT" * 100, b""] + mutated_reads(rng, ref, 200, 100, 200)
    if chunk_bases == 1:
        reads = reads[:120]
    bases, offs = pack(reads)
    exp_out, exp_st = gpu.query_pml_packed(bases, offs)                       # pageable: synchronous path
    exp_z, exp_zst = gpu.query_zml_packed(bases, offs)
    exp_m, exp_c, exp_cst = gpu.query_count_packed(bases, offs)
    exp_cls = gpu.classify_packed(bases, offs, 150, 8)
    ora, ff, sc = cpu.pml_batch(bases, offs, threads=4)
    assert (exp_out == ora).all()
    pb = _pinned_copy(bases)
    gpu.set_option("pipe_chunk_bases", chunk_bases)
    try:
        for rep in range(2):                                                  # second pass: staging reused
            out = movi_amd.pinned_empty(bases.size, np.uint16)
            out[:] = 0xABCD
            got, st, err, rc = gpu.query_pml_packed(pb, offs, want_err=True, out=out)
            assert rc == 0 and got is out and (out == exp_out).all() and not err.any()
            assert (st.bases, st.fast_forwards, st.scans, st.repositions, st.errors) == \
                   (exp_st.bases, exp_st.fast_forwards, exp_st.scans, exp_st.repositions, 0)
            assert (st.fast_forwards, st.scans) == (ff, sc)
            zout = movi_amd.pinned_empty(bases.size, np.uint16)
            zout[:] = 0xABCD
            _, zst = gpu.query_zml_packed(pb, offs, out=zout)
            assert (zout == exp_z).all() and (zst.fast_forwards, zst.scans) == (exp_zst.fast_forwards, exp_zst.scans)
            m, c, cst = gpu.query_count_packed(pb, offs)                      # per-read results stay pageable
            assert (m == exp_m).all() and (c == exp_c).all()
            assert (cst.fast_forwards, cst.scans) == (exp_cst.fast_forwards, exp_cst.scans)
            cls = gpu.classify_packed(pb, offs, 150, 8)
            assert all((x == y).all() for x, y in zip(cls, exp_cls))
    finally:
        gpu.set_option("pipe_chunk_bases", 0)
    # page-locked reads with a pageable result vector: PML / ZML fall back to the synchronous path, same answers
    out2, _ = gpu.query_pml_packed(pb, offs)
    assert (out2 == exp_out).all()
    # "host_overlap" 0 (round 5; what `movi query` sets): page-locked buffers, the call kept whole -- one direct upload, the walk, one download
    gpu.set_option("host_overlap", 0)
    try:
        out3 = movi_amd.pinned_empty(bases.size, np.uint16)
        out3[:] = 0xABCD
        _, st3 = gpu.query_pml_packed(pb, offs, out=out3)
        assert (out3 == exp_out).all() and (st3.fast_forwards, st3.scans) == (ff, sc)
        m3, c3, _ = gpu.query_count_packed(pb, offs)
        assert (m3 == exp_m).all() and (c3 == exp_c).all()
        with pytest.raises(movi_amd.MoviError):
            gpu.set_option("host_overlap", 2)
    finally:
        gpu.set_option("host_overlap", 1)


def test_overlapped_host_path_reports_invariant_violations(built_lib, golden_image):
    """Error bytes and MOVI_ERR_INVARIANT come back from chunks in flight exactly as from the synchronous path."""
    import movi_amd
    img = bytearray(golden_image(6))
    _, _, off, _ = movi_amd.parse_index_image(bytes(img))
    rows = np.frombuffer(img, np.uint8, count=118209 * 8, offset=off).reshape(-1, 8).copy()
    rows[:, 0:4] = 0xFF                                   # every destination id >= r
    img[off: off + rows.size] = rows.tobytes()
    gpu = movi_amd.MoveIndex.from_image(bytes(img))
    bases, offs = pack([b"ACGTACGT", b"A", b"", b"GG", b"T"] * 7)
    exp, est, eerr, erc = gpu.query_pml_packed(bases, offs, want_err=True)
    assert erc == -6 and est.errors == 14 and list(eerr) == [1, 0, 0, 1, 0] * 7
    gpu.set_option("pipe_chunk_bases", 9)
    pb = _pinned_copy(bases)
    out = movi_amd.pinned_empty(bases.size, np.uint16)
    _, st, err, rc = gpu.query_pml_packed(pb, offs, want_err=True, out=out)
    assert rc == -6 and st.errors == 14 and list(err) == list(eerr) and (out == exp).all()


def test_overlapped_host_path_large(built_lib):
    """1 M x 150 bp through the overlapped path with its own chunk policy (several chunks of >= 2^25 bases)."""
    import movi_amd
    from tools import synth
    six = synth.synth_index(2_000_000, mode=6, seed=5)
    gpu = movi_amd.MoveIndex.from_image(six.image())
    bases, offs = synth.synth_reads(six, 1_000_000, 150, seed=6, sub_rate=0.01, n_rate=0.001)
    exp, est = gpu.query_pml_packed(bases, offs)
    pb, out = _pinned_copy(bases), movi_amd.pinned_empty(bases.size, np.uint16)
    got, st = gpu.query_pml_packed(pb, offs, out=out)
    assert (out == exp).all()
    assert (st.bases, st.fast_forwards, st.scans, st.repositions) == (est.bases, est.fast_forwards, est.scans, est.repositions)


def test_index_load_maps_the_file(built_lib, golden_image, tmp_path):
    """movi_index_load maps the index file and uploads the rows in pieces from the mapping: a table of several pieces
    (200 MB: four 64 MiB pieces, the last one partial) must arrive byte for byte, and truncated / empty files fail cleanly."""
    import torch
    import movi_amd
    from tools import synth
    six = synth.synth_index(25_000_000, mode=6, seed=77)
    img = six.image()
    d = tmp_path / "idx"
    d.mkdir()
    np.asarray(img).tofile(str(d / "index.movi"))
    ix = movi_amd.MoveIndex.load(str(d))
    _, _, off, nb = movi_amd.parse_index_image(img)
    ptr, n = ix.device_rows()
    assert n == nb == 25_000_000 * 8
    got = torch.empty(n, dtype=torch.uint8, device="cuda")
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(C.c_void_p(got.data_ptr()), C.c_void_p(ptr), C.c_size_t(n), 3) == 0     # device to device
    assert (got.cpu().numpy() == np.frombuffer(img, np.uint8, count=nb, offset=off)).all()
    ix.close()
    (d / "index.movi").write_bytes(bytes(np.asarray(img)[: off + 1000]))
    with pytest.raises(movi_amd.MoviError) as e:
        movi_amd.MoveIndex.load(str(d))
    assert e.value.code == -2
    (d / "index.movi").write_bytes(b"")
    with pytest.raises(movi_amd.MoviError) as e:
        movi_amd.MoveIndex.load(str(d))
    assert e.value.code == -3


# ---------------------------------------------------------------- segment-parallel long reads ("seg_len")

@pytest.mark.parametrize("mode", [6, 8])
@pytest.mark.parametrize("seg_len", [32, 64, 256, 1024])
def test_segment_parallel_pml_vs_oracle(engines, mode, seg_len):
    """Batches of long reads are cut into segments walked by their own lanes and stitched where the walks fall into step
    (K1 / K2 / K3 of movi_kernels.hpp).  PMLs, error bytes and the fast-forward / scan / reposition counters must be
    exactly those of the oracle and of the one-lane-per-read path, whatever the segment length; noisy reads stitch,
    exact substrings mostly do not and go through the walk-again path."""
    from oracle import build_index as B
    gpu, cpu = engines[mode]
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(7000 + mode + seg_len)
    noisy = mutated_reads(rng, ref, 150, max(800, 3 * seg_len), max(3000, 6 * seg_len))
    for trial, reads in enumerate((
            noisy + [b"", b"A", b"ACGT" * 5] + mutated_reads(rng, ref, 40, 1, 300),                  # mixed with short / empty reads
            [bytes(ref[s:s + L]) for s, L in zip(rng.integers(0, len(ref) - 4000, 60), rng.integers(1500, 4000, 60))],   # exact
            [b"N" * 2500, b"ACGT" * 700, bytes(ref[:3000]), b"T" * 4097])):
        bases, offs = pack(reads)
        gpu.set_option("seg_len", 0)
        ref_out, ref_st = gpu.query_pml_packed(bases, offs)
        exp, ff, sc = cpu.pml_batch(bases, offs, threads=4)
        assert (ref_out == exp).all() and ref_st.segments == 0
        gpu.set_option("seg_len", seg_len)
        gpu.set_option("seg_probe", 0)                     # segments whatever the batch looks like
        try:
            for idx64 in (0, 1):
                gpu.set_option("idx64", idx64)
                out, st, err, rc = gpu.query_pml_packed(bases, offs, want_err=True)
                assert rc == 0 and not err.any()
                assert st.segments > len(reads), (trial, st.segments)             # the segmented path really ran
                bad = np.flatnonzero(out != exp)
                assert bad.size == 0, (trial, seg_len, bad[:10], st.segments, st.rewalked)
                assert (st.fast_forwards, st.scans, st.repositions, st.errors) == \
                       (ref_st.fast_forwards, ref_st.scans, ref_st.repositions, 0), (trial, st.rewalked)
                assert (st.fast_forwards, st.scans) == (ff, sc)
                if trial == 0 and seg_len == 1024:
                    assert st.rewalked < len(reads) // 4                           # noisy reads stitch within a segment this long
        finally:
            gpu.set_option("idx64", 0)
            gpu.set_option("seg_len", 2048)
            gpu.set_option("seg_probe", 1)


def test_segment_parallel_default_policy(engines):
    """Default seg_len (2048): a batch of 10 kbp reads takes the segmented path, a batch of short reads does not; the
    overlapped host path carries its own segment workspace per chunk in flight."""
    import movi_amd
    from oracle import build_index as B
    gpu, cpu = engines[6]
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(7100)
    reads = mutated_reads(rng, ref, 64, 9000, 12000)
    bases, offs = pack(reads)
    out, st = gpu.query_pml_packed(bases, offs)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=4)
    assert (out == exp).all() and (st.fast_forwards, st.scans) == (ff, sc)
    assert st.segments >= 4 * len(reads)
    sb, so = pack(mutated_reads(rng, ref, 300, 100, 200))
    _, st2 = gpu.query_pml_packed(sb, so)
    assert st2.segments == 0
    pb = movi_amd.pinned_empty(bases.size, np.uint8)
    pb[:] = bases
    po = movi_amd.pinned_empty(bases.size, np.uint16)
    gpu.set_option("pipe_chunk_bases", 100_000)
    try:
        _, st3 = gpu.query_pml_packed(pb, offs, out=po)
    finally:
        gpu.set_option("pipe_chunk_bases", 0)
    assert (po == exp).all() and (st3.fast_forwards, st3.scans) == (ff, sc) and st3.segments >= 4 * len(reads)


def test_segment_parallel_reports_invariant_violations(built_lib, golden_image):
    """A corrupted table under the segmented path: the reads are walked again end to end and flagged exactly as without it."""
    import movi_amd
    from oracle import build_index as B
    img = bytearray(golden_image(6))
    _, _, off, _ = movi_amd.parse_index_image(bytes(img))
    rows = np.frombuffer(img, np.uint8, count=118209 * 8, offset=off).reshape(-1, 8).copy()
    rows[1000:60000, 0:4] = 0xFF                          # many destination ids >= r
    img[off: off + rows.size] = rows.tobytes()
    gpu = movi_amd.MoveIndex.from_image(bytes(img))
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(7200)
    bases, offs = pack(mutated_reads(rng, ref, 80, 500, 1500))
    gpu.set_option("seg_len", 0)
    exp, est, eerr, erc = gpu.query_pml_packed(bases, offs, want_err=True)
    assert erc == -6 and est.errors > 0
    gpu.set_option("seg_len", 64)
    gpu.set_option("seg_probe", 0)
    out, st, err, rc = gpu.query_pml_packed(bases, offs, want_err=True)
    assert rc == -6 and st.segments > 80
    assert (err == eerr).all() and (out == exp).all() and st.errors == est.errors


def test_segment_parallel_probe_decides(engines):
    """The probe (two speculative walks per sampled read, started 128 bases apart): noisy long reads fall into step
    quickly and are cut into segments; exact substrings of the reference have no mismatch to meet at, and a batch of those
    stays on one lane per read.  Either way the answers are the oracle's."""
    from oracle import build_index as B
    gpu, cpu = engines[6]
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(7300)
    noisy = mutated_reads(rng, ref, 100, 5000, 8000)
    exact = [bytes(ref[s:s + L]) for s, L in zip(rng.integers(0, len(ref) - 8000, 100), rng.integers(5000, 8000, 100))]
    for reads, want_segments in ((noisy, True), (exact, False)):
        bases, offs = pack(reads)
        out, st = gpu.query_pml_packed(bases, offs)
        exp, ff, sc = cpu.pml_batch(bases, offs, threads=4)
        assert (out == exp).all() and (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0)
        assert (st.segments > 0) == want_segments, (want_segments, st.segments, st.rewalked)


def test_segment_parallel_with_classification_bins(engines):
    """--classify on a batch that is walked segment-parallel: the bins are reduced from the resident PML vector after the
    walk instead of inside it (a bin spans segments); verdict-only calls keep the vector in the workspace.  Same bins as
    the fused one-lane-per-read kernels, with and without the caller's PML vector."""
    import torch
    from oracle import build_index as B
    gpu, cpu = engines[6]
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(7400)
    reads = mutated_reads(rng, ref, 120, 600, 2500) + [b"", b"ACGT" * 40, bytes(ref[:1000])]
    bases, offs = pack(reads)
    n = len(reads)
    gpu.set_option("seg_len", 0)
    exp_bins = gpu.classify_packed(bases, offs, 150, 8)
    exp_pml, _, _ = cpu.pml_batch(bases, offs, threads=4)
    dev = torch.device("cuda", 0)
    d_bases = torch.from_numpy(bases.copy()).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64).copy()).to(dev)
    gpu.set_option("seg_len", 64)
    gpu.set_option("seg_probe", 0)
    try:
        got = gpu.classify_packed(bases, offs, 150, 8)                       # verdicts only (movi_pml_classify_host)
        assert all((x == y).all() for x, y in zip(got, exp_bins))
        for with_vector in (True, False):
            d_out = torch.zeros(bases.size, dtype=torch.int16, device=dev)
            d_a = torch.full((n,), -1, dtype=torch.int32, device=dev)
            d_b = torch.full((n,), -1, dtype=torch.int32, device=dev)
            d_s = torch.full((n,), -1, dtype=torch.int64, device=dev)
            gpu.pml_classify_device(d_bases.data_ptr(), d_offs.data_ptr(), n, bases.size, 150, 8,
                                    d_out.data_ptr() if with_vector else 0, d_a.data_ptr(), d_b.data_ptr(), d_s.data_ptr())
            torch.cuda.synchronize()
            st = gpu.last_stats()
            assert st.segments > n
            assert (d_a.cpu().numpy().view(np.uint32) == exp_bins[0]).all() and (d_b.cpu().numpy().view(np.uint32) == exp_bins[1]).all()
            assert (d_s.cpu().numpy().view(np.uint64) == exp_bins[2]).all()
            if with_vector:
                assert (d_out.cpu().numpy().view(np.uint16) == exp_pml).all()
            else:
                assert int(d_out.abs().sum().item()) == 0
    finally:
        gpu.set_option("seg_len", 2048)
        gpu.set_option("seg_probe", 1)


@pytest.mark.parametrize("mode", [6, 8, 3])
@pytest.mark.parametrize("seg_len", [32, 64, 1024])
def test_segment_parallel_zml_vs_oracle(engines, built_lib, mode, seg_len):
    """The ZML parse of long reads cut into segments (zml_kernel<.., 1> / zml_stitch_kernel / seg_finalize_kernel): values,
    error bytes and the fast-forward / scan counters equal the oracle's and the one-lane-per-read kernels', on the
    thresholds types and on a threshold-less `regular` index."""
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    if mode == 3:
        img = B.build_index_from_seqs([ref], 3)
        gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    else:
        gpu, cpu = engines[mode]
    rng = np.random.default_rng(7500 + mode + seg_len)
    noisy = mutated_reads(rng, ref, 120, max(800, 3 * seg_len), max(3000, 6 * seg_len))
    for trial, reads in enumerate((
            noisy + [b"", b"A", b"ACGT" * 5] + mutated_reads(rng, ref, 40, 1, 300),
            [bytes(ref[s:s + L]) for s, L in zip(rng.integers(0, len(ref) - 4000, 40), rng.integers(1500, 4000, 40))],
            [b"N" * 2500, b"ACGT" * 700, bytes(ref[:3000]), b"T" * 4097, b"ACGTN" * 500])):
        bases, offs = pack(reads)
        exp = cpu.zml_batch(bases, offs, threads=4)
        gpu.set_option("seg_len", 0)
        ref_out, ref_st = gpu.query_zml_packed(bases, offs)
        assert (ref_out == exp).all() and ref_st.segments == 0
        gpu.set_option("seg_len", seg_len)
        gpu.set_option("seg_probe", 0)
        try:
            for idx64 in (0, 1):
                gpu.set_option("idx64", idx64)
                out, st = gpu.query_zml_packed(bases, offs)
                assert st.segments > len(reads), (trial, st.segments)
                bad = np.flatnonzero(out != exp)
                assert bad.size == 0, (trial, seg_len, idx64, bad[:10], st.segments, st.rewalked)
                assert (st.fast_forwards, st.scans, st.errors) == (ref_st.fast_forwards, ref_st.scans, 0), (trial, st.rewalked)
        finally:
            gpu.set_option("idx64", 0)
            gpu.set_option("seg_len", 2048)
            gpu.set_option("seg_probe", 1)
    # default policy with the probe: noisy 10 kbp reads are cut, exact ones are not; same values either way
    long_noisy = mutated_reads(rng, ref, 64, 9000, 12000)
    long_exact = [bytes(ref[s:s + 9000]) for s in rng.integers(0, len(ref) - 9000, 64)]
    for reads, want in ((long_noisy, True), (long_exact, False)):
        bases, offs = pack(reads)
        out, st = gpu.query_zml_packed(bases, offs)
        assert (out == cpu.zml_batch(bases, offs, threads=4)).all()
        assert (st.segments > 0) == want, (want, st.segments)
    if mode == 3:
        gpu.close()


@pytest.mark.parametrize("seg_len", [32, 64, 96])
def test_segment_parallel_length_edges(engines, seg_len):
    """Read lengths around every boundary of the segment plan (2 S - 1, 2 S, 2 S + 1, multiples of 32 +- 1, one base more
    than a whole number of segments ...), PML and ZML, cut against uncut on the GPU and against the oracle."""
    from oracle import build_index as B
    gpu, cpu = engines[6]
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(7600 + seg_len)
    lens = sorted(set([2 * seg_len + d for d in (-1, 0, 1, 31, 32, 33)] + [k * seg_len + d for k in (3, 4, 7) for d in (-33, -32, -31, -1, 0, 1, 31, 32, 33)] +
                      [5 * seg_len + 17, 1000, 1023, 1024, 1025, 2047, 2049]))
    reads = []
    for L in lens:
        for _ in range(3):
            s = int(rng.integers(0, len(ref) - L))
            r = bytearray(ref[s:s + L])
            for k in np.flatnonzero(rng.random(L) < 0.06):
                r[k] = b"ACGTN"[int(rng.integers(0, 5))]
            reads.append(bytes(r))
    bases, offs = pack(reads)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=4)
    zexp = cpu.zml_batch(bases, offs, threads=4)
    gpu.set_option("seg_len", seg_len)
    gpu.set_option("seg_probe", 0)
    try:
        out, st = gpu.query_pml_packed(bases, offs)
        bad = np.flatnonzero(out != exp)
        assert st.segments > len(reads) and bad.size == 0, (bad[:10], st.segments, st.rewalked)
        assert (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0)
        zout, zst = gpu.query_zml_packed(bases, offs)
        zbad = np.flatnonzero(zout != zexp)
        assert zst.segments > len(reads) and zbad.size == 0, (zbad[:10], zst.segments, zst.rewalked)
    finally:
        gpu.set_option("seg_len", 2048)
        gpu.set_option("seg_probe", 1)
