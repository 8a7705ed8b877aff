"""GPU suite (-m gpu): look-ahead rows ("ahead_rows" option) -- the table's second copy in which every row carries what
its LF target looks like, so that a base that matches there without a fast-forward is resolved without fetching the
target: two bases per gather (reference semantics of the step: src/read_processor.cpp:188-238 match branch + LF_move,
src/move_structure.cpp:59-87).  Only staged short-read launches walk on them.  PMLs, error bytes, bins and the
fast-forward / scan / reposition counters must equal the oracle's and those of the same launch without the copy: on every
index type the PML walk serves, for every read length the staging area holds, next to wavefronts whose reads roll through it, with
illegal bases, separators, corrupt rows, both row-index widths and tables whose last window sits differently in its line."""
import numpy as np
import pytest

from test_gpu_parity import mutated_reads, pack
from test_top_of_walk_gpu import _ref

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _vectors_straight_from_the_walk(packer_paths):
    """This module is about the look-ahead rows AND the walk's own output paths (register packer, LDS ring), asserted by kernel name:
    its handles write the vector themselves ("pml_via_mask" 0).  The mask output on every layout: tests/test_mask_gpu.py,
    tests/test_deep_rows_gpu.py, tests/test_kernel_coverage_gpu.py."""
    yield

CAP = 336          # bases per lane the default occupancy cap's LDS padding holds (7 wavefronts per CU) ...
CAP_AHEAD = 256    # ... and the cap on the look-ahead rows (9 wavefronts per CU)
N_BIG = 300_000    # > 256 CUs x 64 lanes x 18 wavefronts: the capped, staged launch


def _big_batch(ref, rng, n=N_BIG, max_len=CAP, sub=0.02, n_long=40, alphabet=b"ACGT"):
    lens = rng.integers(0, max_len + 1, n).astype(np.uint64)
    if n_long:
        lens[rng.choice(n, n_long, replace=False)] = rng.integers(CAP + 1, 2000, n_long)   # wavefronts that roll through their staged stretch
    lens[:64] = CAP
    lens[64:128] = 0
    lens[128:192] = 1
    lens[192:256] = 2
    starts = rng.integers(0, len(ref) - 2000, n)
    offs = np.zeros(n + 1, np.uint64)
    np.cumsum(lens, out=offs[1:])
    refa = np.frombuffer(ref, np.uint8)
    idx = np.repeat(starts.astype(np.int64) - offs[:-1].astype(np.int64), lens.astype(np.int64)) + np.arange(int(offs[-1]), dtype=np.int64)
    bases = refa[idx].copy()
    mut = rng.random(bases.size)
    bases[mut < sub] = np.frombuffer(alphabet, np.uint8)[rng.integers(0, len(alphabet), int((mut < sub).sum()))]
    bases[(mut >= sub) & (mut < sub + 0.003)] = ord("N")
    return bases, offs


@pytest.mark.parametrize("mode", [6, 8, 7])
def test_ahead_rows_vs_oracle(built_lib, golden_image, mode):
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = _ref()
    img = golden_image(mode) if mode != 7 else B.build_index_from_seqs([ref], 7)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    bases, offs = _big_batch(ref, np.random.default_rng(9100 + mode))
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=8)
    gpu.set_option("ahead_rows", 0)
    gpu.set_option("kmer_k", 0)
    out0, st0 = gpu.query_pml_packed(bases, offs)
    li = gpu.last_launch()
    assert li["ahead"] == 0 and li["staged"] == CAP and li["kernel"].endswith(", 0, 1, 0, 0, 0>")
    assert (out0 == exp).all() and (st0.fast_forwards, st0.scans, st0.errors) == (ff, sc, 0)
    bins0 = gpu.classify_packed(bases, offs, 40, 4)
    gpu.set_option("ahead_rows", 1)
    for K in (0, 12):
        gpu.set_option("kmer_k", K)
        out, st = gpu.query_pml_packed(bases, offs)
        li = gpu.last_launch()
        assert li["ahead"] == 1 and (li["staged"], li["waves_per_cu"]) == (CAP_AHEAD, 9) and li["kernel"].endswith(", 0, 1, 1, 0, 0>")
        assert (out == exp).all(), (mode, K)
        assert (st.fast_forwards, st.scans, st.repositions, st.errors) == (ff, sc, st0.repositions, 0), (mode, K)
        assert st.lane_steps < 0.8 * st0.lane_steps, (mode, K)          # the point of it: most bases ride along
        bins = gpu.classify_packed(bases, offs, 40, 4)                  # fused bins with the vector ...
        assert all((x == y).all() for x, y in zip(bins, bins0)), K
    # ... a small batch of long reads: uncapped, rolling through its staged stretch many times
    small = mutated_reads(np.random.default_rng(9200), ref, 300, 1, 3000)
    sb, so = pack(small)
    sexp, sff, ssc = cpu.pml_batch(sb, so, threads=4)
    gpu.set_option("seg_len", 0)                                        # (one lane per read: not the segment-parallel plan)
    sout, sst = gpu.query_pml_packed(sb, so)
    li = gpu.last_launch()
    assert li["ahead"] == 1 and li["waves_per_cu"] == 0 and li["staged"] == CAP and li["segmented"] == 0
    assert (sout == sexp).all() and (sst.fast_forwards, sst.scans) == (sff, ssc)
    # ... and batches between one and 18 wavefronts per CU, which get what their wavefronts leave of the CU's LDS
    for n_mid, cap_mid in ((150_000, 192), (280_000, 96)):
        mb, mo = bases[: int(offs[n_mid])], offs[: n_mid + 1]
        mout, mst = gpu.query_pml_packed(mb, mo)
        li = gpu.last_launch()
        assert li["ahead"] == 1 and li["waves_per_cu"] == 0 and li["staged"] == cap_mid, li
        assert (mout == exp[: mb.size]).all() and mst.errors == 0
    gpu.set_option("seg_len", 2048)
    # 64-bit row indexes: the same walk on the other instantiation
    gpu.set_option("idx64", 1)
    out, st = gpu.query_pml_packed(bases, offs)
    assert gpu.last_launch()["ahead"] == 1 and gpu.last_launch()["idx64"] == 1
    assert (out == exp).all() and (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0)
    gpu.close()
    cpu.close()


def test_ahead_rows_are_built_by_the_first_pml_query_on_a_small_table(built_lib, golden_image):
    import movi_amd
    from oracle.oracle import Oracle
    img = golden_image(6)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    bases, offs = _big_batch(_ref(), np.random.default_rng(9300), n_long=0, max_len=200)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=8)
    out, st = gpu.query_pml_packed(bases, offs)                         # nothing set: table + look-ahead rows (+ round 6: deep rows) by themselves
    assert gpu.last_launch()["ahead"] == 2 and gpu.info("ahead_rows_bytes") > 0 and gpu.info("deep_rows_bytes") > 0   # short reads on a small table of real text: the deep rows
    assert (out == exp).all() and (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0)
    gpu.set_option("deep", 0)                                           # ... without them: the look-ahead rows the same query built
    out, st = gpu.query_pml_packed(bases, offs)
    assert gpu.last_launch()["ahead"] == 1
    assert (out == exp).all() and (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0)
    m, c, _ = gpu.query_count_packed(bases[: int(offs[1000])], offs[:1001])   # other queries are not affected
    em, ec = cpu.count_batch(bases[: int(offs[1000])], offs[:1001], threads=4)
    assert (m == em).all() and (c == ec).all()
    gpu.set_option("ahead_rows", 0)
    out, st2 = gpu.query_pml_packed(bases, offs)
    assert gpu.last_launch()["ahead"] == 0 and (out == exp).all() and st2.lane_steps > st.lane_steps
    gpu.close()
    cpu.close()


@pytest.mark.parametrize("cut", [0, 1, 2, 3, 4, 5, 6, 7])
def test_ahead_rows_last_window(built_lib, cut):
    """The walk's last window is pulled back to rows r-4 .. r-1 and has a line of its own in the copy: tables whose row count
    leaves every remainder modulo 8 (every read starts in that window, at row r-1)."""
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = _ref()
    extra = {1: 0, 5: 7, 0: 14, 2: 28, 6: 42, 3: 49, 7: 105, 4: 112}[cut]     # prefixes of the reference whose tables have r % 8 == cut
    img = B.build_index_from_seqs([ref[: 30000 + extra]], 6)
    assert movi_amd.parse_index_image(img)[1].r % 8 == cut
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    bases, offs = _big_batch(ref[:30000], np.random.default_rng(9400 + cut), max_len=64, n_long=3)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=8)
    gpu.set_option("ahead_rows", 1)
    for K in (0, 5):
        gpu.set_option("kmer_k", K)
        out, st = gpu.query_pml_packed(bases, offs)
        assert gpu.last_launch()["ahead"] == 1
        assert (out == exp).all() and (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0), (cut, K)
    gpu.close()
    cpu.close()


def test_ahead_rows_on_a_separators_index(built_lib):
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = _ref()
    img = B.build_index_from_seqs([ref[:40000], ref[40000:90000], ref[90000:]], 6, separators=True)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    bases, offs = _big_batch(ref, np.random.default_rng(9500), alphabet=b"ACGT%")
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=8)
    gpu.set_option("ahead_rows", 1)
    for K in (0, 9):
        gpu.set_option("kmer_k", K)
        out, st = gpu.query_pml_packed(bases, offs)
        assert gpu.last_launch()["ahead"] == 1 and gpu.last_launch()["kernel"].endswith(", 0, 1, 0, 1, 1, 0, 0>")
        assert (out == exp).all() and (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0), K
    gpu.close()
    cpu.close()


def test_ahead_rows_with_corrupt_rows(built_lib, golden_image):
    """Rows whose id (or whose target's id) points past the table carry an invalid entry: those steps are taken one by one
    and run into the reference's throw exactly as without the copy."""
    import movi_amd
    img = bytearray(golden_image(6))
    _, _, off, _ = movi_amd.parse_index_image(bytes(img))
    rows = np.frombuffer(img, np.uint8, count=118209 * 8, offset=off).reshape(-1, 8).copy()
    rng = np.random.default_rng(9600)
    hit = rng.choice(118209, 3000, replace=False)
    rows[hit, 0:4] = 0xFF
    img[off: off + rows.size] = rows.tobytes()
    gpu = movi_amd.MoveIndex.from_image(bytes(img))
    bases, offs = _big_batch(_ref(), rng, max_len=120)
    gpu.set_option("ahead_rows", 0)
    e_out, e_st, e_err, e_rc = gpu.query_pml_packed(bases, offs, want_err=True)
    assert e_rc == -6 and e_st.errors > 1000
    gpu.set_option("ahead_rows", 1)
    for K in (0, 8):
        gpu.set_option("kmer_k", K)
        out, st, err, rc = gpu.query_pml_packed(bases, offs, want_err=True)
        assert gpu.last_launch()["ahead"] == 1
        assert rc == -6 and (out == e_out).all() and (err == e_err).all() and st.errors == e_st.errors, K
    gpu.close()


def test_ahead_rows_refused_where_they_cannot_serve(built_lib):
    import movi_amd
    from oracle import build_index as B
    gpu = movi_amd.MoveIndex.from_image(B.build_index_from_seqs([_ref()[:20000]], 3))   # no thresholds: no PML walk
    with pytest.raises(movi_amd.MoviError):
        gpu.set_option("ahead_rows", 1)
    with pytest.raises(movi_amd.MoviError):
        gpu.set_option("ahead_rows", 2)                  # (round 4's chain rows: measured slower, removed)
    gpu.close()


@pytest.mark.parametrize("sep", [0, 1])
def test_segments_walk_on_the_look_ahead_rows(built_lib, golden_image, sep):
    """Segment-parallel long reads: K1's lanes (one segment each, a checkpoint every 32 bases and the segment's final state
    recorded -- also when the base in question is the second of a two-base step) and K3's (whole reads again) stage their
    bases through LDS and walk on the look-ahead rows like any launch.  PMLs, error bytes and counters equal the oracle's
    for segment lengths that put checkpoints and segment ends on both bases of such steps."""
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = _ref()
    img = B.build_index_from_seqs([ref[:60000], ref[60000:]], 6, separators=True) if sep else golden_image(6)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    rng = np.random.default_rng(9700 + sep)
    reads = mutated_reads(rng, ref, 300, 900, 6000) + [bytes(ref[1000:9000]), bytes(ref[20000:20000 + 4097])]   # clean reads: long runs of two-base steps
    bases, offs = pack(reads)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=8)
    gpu.set_option("seg_probe", 0)
    for seg_len in (32, 64, 96, 512):
        gpu.set_option("seg_len", seg_len)
        for ahead in (1, 0):
            gpu.set_option("ahead_rows", ahead)
            for stage in (1, 0):
                gpu.set_option("stage_reads", stage)
                out, st = gpu.query_pml_packed(bases, offs)
                li = gpu.last_launch()
                assert li["segmented"] == 1 and li["ahead"] == (ahead & stage) and (li["staged"] > 0) == bool(stage), (seg_len, li)
                # (staged segments: their PMLs leave through the LDS ring -- RING = 1, the last template argument)
                assert li["kernel"].endswith(", 1, %d, %d, 0, %d>" % (stage, ahead & stage, stage))   # SEG, STG, AHD, PSH, RING
                assert (out == exp).all(), (seg_len, ahead, stage)
                assert (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0), (seg_len, ahead, stage)
                assert st.segments > len(reads)
    gpu.close()
    cpu.close()


@pytest.mark.parametrize("mode", [6, 8, 7])
def test_count_on_the_look_ahead_rows_vs_oracle(built_lib, golden_image, mode):
    """The count query (backward search, src/move_structure_search.cpp:169-352) on the look-ahead rows: when both ends' LF
    targets hold the base after the current one and neither fast-forwards there, that base's whole step comes out of the
    entries.  matched / count, error bytes and the fast-forward / scan counters equal the oracle's and the plain-rows
    kernel's: reads that match to their first base, reads that stop early, illegal bases next to and inside two-base steps,
    with and without the interval table."""
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = _ref()
    img = golden_image(mode) if mode != 7 else B.build_index_from_seqs([ref], 7)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    rng = np.random.default_rng(9800 + mode)
    reads = mutated_reads(rng, ref, 3000, 1, 400)
    reads += [bytes(ref[s: s + L]) for s, L in ((100, 1), (100, 2), (100, 3), (5000, 150), (7000, 1000), (9000, 4000))]   # exact substrings: two-base steps all the way
    base = bytearray(ref[30000:30200])
    for pos in (1, 2, 3, 4, 13, 14, 15):                                 # an illegal base at distance pos from the read's end
        for bad in (b"N", b"a"):
            r = bytearray(base)
            r[len(r) - pos: len(r) - pos + 1] = bad
            reads.append(bytes(r))
    reads += [b"", b"A", b"ACGT" * 40, bytes(rng.choice(list(b"ACGT"), size=60).astype(np.uint8))]
    bases, offs = pack(reads)
    em, ec = cpu.count_batch(bases, offs, threads=8)
    gpu.set_option("count_variant", 0)                                   # count_kernel_v0 (round 5: the default is the lane state machine on the plain rows)
    gpu.set_option("ahead_rows", 0)
    for K in (0, 12):
        gpu.set_option("ftab_k", K)
        gpu.set_option("ahead_rows", 0)
        m0, c0, st0 = gpu.query_count_packed(bases, offs)
        assert gpu.last_launch()["ahead"] == 0 and gpu.last_launch()["kernel"] == "count_kernel_v0<6, 0>"
        assert (m0 == em).all() and (c0 == ec).all()
        gpu.set_option("ahead_rows", 1)
        m, c, st = gpu.query_count_packed(bases, offs)
        assert gpu.last_launch()["ahead"] == 1 and gpu.last_launch()["kernel"] == "count_kernel_v0<6, 1>"
        assert (m == em).all() and (c == ec).all(), (mode, K)
        assert (st.fast_forwards, st.scans, st.errors) == (st0.fast_forwards, st0.scans, 0), (mode, K)
    # a big batch (the capped launch)
    bb, bo = _big_batch(ref, rng, max_len=200, n_long=20)
    em, ec = cpu.count_batch(bb, bo, threads=8)
    m, c, st = gpu.query_count_packed(bb, bo)
    assert gpu.last_launch()["ahead"] == 1 and (m == em).all() and (c == ec).all() and st.errors == 0
    gpu.close()
    cpu.close()


@pytest.mark.parametrize("mode", [6, 8, 7])
def test_zml_on_the_look_ahead_rows_vs_oracle(built_lib, golden_image, mode):
    """The ZML parse (query_zml, src/move_structure_query.cpp:690-785) as a lane state machine on the look-ahead rows: a base
    both of whose LF moves land without a fast-forward -- known from the entries of the interval's two ends -- is complete
    without the target rows, and the next base starts from what the entries say about them (zml_kernel_flat<6, T, 0, 1>).
    Match lengths, error bytes and the fast-forward / scan counters equal the oracle's and the plain-rows kernel's: exact
    substrings (look-ahead steps all the way), reads with substitutions and illegal bases next to and inside such steps,
    every read length around the 8- / 16-base groups of the packed I/O, both index widths, small and big batches."""
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = _ref()
    img = golden_image(mode) if mode != 7 else B.build_index_from_seqs([ref], 7)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    rng = np.random.default_rng(9900 + mode)
    reads = mutated_reads(rng, ref, 2500, 1, 400)
    reads += [bytes(ref[s: s + L]) for s, L in ((100, 1), (100, 2), (100, 3), (200, 7), (200, 8), (200, 9), (300, 15), (300, 16), (300, 17),
                                                (400, 31), (400, 32), (400, 33), (5000, 150), (7000, 1000), (9000, 4000))]
    base = bytearray(ref[30000:30200])
    for pos in (1, 2, 3, 4, 5, 13, 14, 15, 16, 17, 18):                  # an illegal base at distance pos from the read's end
        for bad in (b"N", b"a"):
            r = bytearray(base)
            r[len(r) - pos: len(r) - pos + 1] = bad
            reads.append(bytes(r))
    reads += [b"", b"A", b"N", b"ACGT" * 40, bytes(rng.choice(list(b"ACGT"), size=60).astype(np.uint8))]
    bases, offs = pack(reads)
    exp = cpu.zml_batch(bases, offs, threads=8)
    gpu.set_option("seg_len", 0)
    gpu.set_option("zml_variant", 1)
    gpu.set_option("zml_ahead", 1)                                        # (opt-in: a third fewer iterations, no faster)
    for idx64 in (0, 1):
        gpu.set_option("idx64", idx64)
        gpu.set_option("ahead_rows", 0)
        out0, st0 = gpu.query_zml_packed(bases, offs)
        li = gpu.last_launch()
        assert li["ahead"] == 0 and li["kernel"].startswith("zml_kernel_flat<6, ") and li["kernel"].endswith(", 0>")
        assert (out0 == exp).all() and st0.errors == 0
        gpu.set_option("ahead_rows", 1)
        out, st = gpu.query_zml_packed(bases, offs)
        li = gpu.last_launch()
        assert li["ahead"] == 1 and li["kernel"].endswith(", 0, 1, 0, 0>") and li["idx64"] == idx64
        assert (out == exp).all(), (mode, idx64)
        assert (st.fast_forwards, st.scans, st.errors) == (st0.fast_forwards, st0.scans, 0), (mode, idx64)
    gpu.set_option("idx64", 0)
    bb, bo = _big_batch(ref, rng, max_len=200, n_long=20)
    bexp = cpu.zml_batch(bb, bo, threads=8)
    bout, bst = gpu.query_zml_packed(bb, bo)
    assert gpu.last_launch()["ahead"] == 1 and (bout == bexp).all() and bst.errors == 0
    gpu.close()
    cpu.close()


def test_count_on_the_look_ahead_rows_separators_and_corrupt_rows(built_lib, golden_image):
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = _ref()
    img = B.build_index_from_seqs([ref[:40000], ref[40000:90000], ref[90000:]], 6, separators=True)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    rng = np.random.default_rng(9900)
    reads = mutated_reads(rng, ref, 2000, 1, 300) + [bytes(ref[39950:40050]), b"ACG%TACGTACGTACGT", bytes(ref[100:1100])]
    bases, offs = pack(reads)
    em, ec = cpu.count_batch(bases, offs, threads=8)
    gpu.set_option("count_variant", 0)
    for ahead in (0, 1):
        gpu.set_option("ahead_rows", ahead)
        m, c, st = gpu.query_count_packed(bases, offs)
        assert gpu.last_launch()["ahead"] == ahead and (m == em).all() and (c == ec).all() and st.errors == 0, ahead
    gpu.close()
    cpu.close()
    # rows pointing past the table: the entries of such rows (and of rows that point AT them) are invalid, the search runs
    # into the reference's throw exactly where the plain-rows kernel does
    img = bytearray(golden_image(6))
    _, _, off, _ = movi_amd.parse_index_image(bytes(img))
    rows = np.frombuffer(img, np.uint8, count=118209 * 8, offset=off).reshape(-1, 8).copy()
    rows[rng.choice(118209, 3000, replace=False), 0:4] = 0xFF
    img[off: off + rows.size] = rows.tobytes()
    gpu = movi_amd.MoveIndex.from_image(bytes(img))
    gpu.set_option("count_variant", 0)
    gpu.set_option("ahead_rows", 0)
    m0, c0, st0, err0, rc0 = gpu.query_count_packed(bases, offs, want_err=True)
    assert st0.errors > 100
    gpu.set_option("ahead_rows", 1)
    m, c, st, err, rc = gpu.query_count_packed(bases, offs, want_err=True)
    assert gpu.last_launch()["ahead"] == 1
    assert rc == rc0 and (m == m0).all() and (c == c0).all() and (err == err0).all() and st.errors == st0.errors
    gpu.close()


@pytest.mark.parametrize("kind", ["repeats", "poly", "two_letters", "random", "tandem", "with_n_runs"])
def test_look_ahead_on_odd_texts(built_lib, kind):
    """Texts that stress what the entries encode: long runs (offsets near the 11-bit limit, rows split at 2047), two-letter
    alphabets (no top-of-walk table), tandem repeats (every base rides along for thousands of steps), random text (almost
    none does).  PML and count against the oracle, small batches (uncapped, staged launches)."""
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    rng = np.random.default_rng(10000 + ["repeats", "poly", "two_letters", "random", "tandem", "with_n_runs"].index(kind))
    acgt = np.frombuffer(b"ACGT", np.uint8)
    if kind == "repeats":
        unit = bytes(acgt[rng.integers(0, 4, 300)])
        text = b"".join(bytes(bytearray(unit)) if rng.random() < 0.7 else bytes(acgt[rng.integers(0, 4, 300)]) for _ in range(120))
    elif kind == "poly":
        text = b"".join((b"A" * int(rng.integers(1, 6000))) + bytes(acgt[rng.integers(0, 4, int(rng.integers(1, 40)))]) for _ in range(12))
    elif kind == "two_letters":
        text = bytes(np.frombuffer(b"AT", np.uint8)[rng.integers(0, 2, 30000)])
    elif kind == "random":
        text = bytes(acgt[rng.integers(0, 4, 40000)])
    elif kind == "tandem":
        text = (b"ACGTTGCA" * 3000) + bytes(acgt[rng.integers(0, 4, 2000)]) + (b"GATTACA" * 2000)
    else:
        text = bytes(acgt[rng.integers(0, 4, 20000)]) + b"N" * 50 + bytes(acgt[rng.integers(0, 4, 20000)])
    text = text.replace(b"N", b"")                                        # the index itself is over ACGT
    img = B.build_index_from_seqs([text], 6)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    reads = mutated_reads(rng, text, 1500, 1, 700) + [bytes(text[:5000]), bytes(text[-3000:]), bytes(text[1000:1001])]
    if kind == "with_n_runs":
        reads += [bytes(text[100:160]) + b"NNN" + bytes(text[160:260]), b"N" * 40, b"ACGTN" * 30]
    bases, offs = pack(reads)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=8)
    em, ec = cpu.count_batch(bases, offs, threads=8)
    gpu.set_option("seg_len", 0)
    for ahead in (0, 1):
        gpu.set_option("ahead_rows", ahead)
        out, st = gpu.query_pml_packed(bases, offs)
        assert gpu.last_launch()["ahead"] == ahead and gpu.last_launch()["staged"] > 0
        assert (out == exp).all(), (kind, ahead)
        assert (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0), (kind, ahead)
        for cv in (0, -1):                                 # count_kernel_v0 (on the copy where it is there) and the default state machine (plain rows)
            gpu.set_option("count_variant", cv)
            m, c, cst = gpu.query_count_packed(bases, offs)
            assert gpu.last_launch()["ahead"] == (1 if ahead == 1 and cv == 0 else 0)
            assert (m == em).all() and (c == ec).all() and cst.errors == 0, (kind, ahead, cv)
    gpu.close()
    cpu.close()


def test_count_uses_the_copy_where_the_table_says_it_pays(built_lib):
    """Built by itself, the look-ahead copy serves the count query only on tables whose positions mostly arrive at their LF
    target without a fast-forward (the builder tallies it: 0.83 on BWTs of real text, 0.51 on the uniformly random run
    sequences of tools/synth.c, where the copy costs the count query 12 %); PML walks use it either way, and a caller who
    asks for the copy gets it for both.  Same answers whichever table is walked."""
    import movi_amd
    from oracle.oracle import Oracle
    from tools import synth
    six = synth.synth_index(300_000, mode=6, seed=77)
    img = six.image()
    bases, offs = synth.synth_reads(six, 4000, 120, seed=78, sub_rate=0.02, n_rate=0.002)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    em, ec = cpu.count_batch(bases, offs, threads=8)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=8)
    out, st = gpu.query_pml_packed(bases, offs)                        # builds the copy (a small table)
    assert gpu.last_launch()["ahead"] == 1 and (out == exp).all() and (st.fast_forwards, st.scans) == (ff, sc)
    m, c, _ = gpu.query_count_packed(bases, offs)                      # the default: the state machine, on the plain rows
    assert gpu.last_launch()["ahead"] == 0 and gpu.last_launch()["kernel"] == "zml_kernel_flat<6, unsigned int, 0, 0, 0, 1>"
    assert (m == em).all() and (c == ec).all()
    gpu.set_option("count_variant", 0)                                 # count_kernel_v0: the copy only where the table says it pays
    m, c, _ = gpu.query_count_packed(bases, offs)
    assert gpu.last_launch()["ahead"] == 0 and gpu.last_launch()["kernel"] == "count_kernel_v0<6, 0>"   # a random table: plain rows
    assert (m == em).all() and (c == ec).all()
    gpu.set_option("ahead_rows", 1)                                   # on request: both
    m, c, _ = gpu.query_count_packed(bases, offs)
    assert gpu.last_launch()["ahead"] == 1 and (m == em).all() and (c == ec).all()
    gpu.close()
    cpu.close()


@pytest.mark.parametrize("mode", [6, 8])
def test_pair_shared_gathers_vs_oracle(built_lib, golden_image, mode):
    """"pair_loads" 1 (pml_kernel_flatp<..., PSH = 1>): the two lanes of a pair fetch their row windows together -- one load
    instruction per lane of the pair, each lane one 16-byte half, halves exchanged across the pair -- on the plain rows and on the
    look-ahead rows.  Same PMLs, error bytes, counters and bins as the oracle and as the lane-private loads: odd numbers of
    reads (a lane whose partner has no read), reads that end while their partner walks on, long reads rolling through the
    staged stretch, both index widths, the table's last window."""
    import movi_amd
    from oracle.oracle import Oracle
    img = golden_image(mode)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    ref = _ref()
    rng = np.random.default_rng(9990 + mode)
    bases, offs = _big_batch(ref, rng)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=8)
    small = mutated_reads(rng, ref, 301, 1, 3000) + [b"", b"A", b"N" * 5, bytes(ref[-200:]), bytes(ref[:200])]
    sb, so = pack(small)
    sexp, sff, ssc = cpu.pml_batch(sb, so, threads=4)
    gpu.set_option("seg_len", 0)
    for ahead in (0, 1):
        gpu.set_option("ahead_rows", ahead)
        gpu.set_option("pair_loads", 0)
        bins0 = gpu.classify_packed(bases, offs, 40, 4)
        gpu.set_option("pair_loads", 1)
        for K in (0, 12):
            gpu.set_option("kmer_k", K)
            for idx64 in (0, 1):
                gpu.set_option("idx64", idx64)
                out, st = gpu.query_pml_packed(bases, offs)
                li = gpu.last_launch()
                assert li["ahead"] == ahead and li["kernel"].endswith(", 0, 1, %d, 1, 0>" % ahead) and li["idx64"] == idx64, li
                assert (out == exp).all(), (mode, ahead, K, idx64)
                assert (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0), (mode, ahead, K, idx64)
            gpu.set_option("idx64", 0)
        bins = gpu.classify_packed(bases, offs, 40, 4)
        assert all((x == y).all() for x, y in zip(bins, bins0)), ahead
        sout, sst = gpu.query_pml_packed(sb, so)
        assert gpu.last_launch()["kernel"].endswith(", 1>") and (sout == sexp).all() and (sst.fast_forwards, sst.scans) == (sff, ssc)
    gpu.set_option("pair_loads", 0)
    gpu.close()
    cpu.close()


@pytest.mark.parametrize("mode", [6, 8])
def test_pml_out_ring_vs_oracle(built_lib, golden_image, mode):
    """"out_ring" (pml_kernel_flatp<..., RING = 1>): the staged PML kernels' PMLs leave through a ring of 32 per lane in LDS -- one 2-byte LDS
    write per PML, a finished group of 16 as two 16-byte stores -- instead of the register packer.  On by itself for batches of
    long reads, "out_ring" 1 wherever the block's LDS holds it.  Same PML vectors, error bytes, counters and bins: reads of every
    length from 0 up (tails of 0 .. 15 PMLs stored one by one, reads shorter than a group), failing reads (zero-filled), long
    reads rolling through the staged stretch, pair-shared gathers, segments."""
    import movi_amd
    from oracle.oracle import Oracle
    img = golden_image(mode)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    ref = _ref()
    rng = np.random.default_rng(9890 + mode)
    bases, offs = _big_batch(ref, rng)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=8)
    long_reads = mutated_reads(rng, ref, 301, 1, 3000) + [bytes(ref[-2500:]), bytes(ref[:2100])]
    lb, lo = pack(long_reads)
    lexp, lff, lsc = cpu.pml_batch(lb, lo, threads=4)
    ragged = [bytes(ref[s: s + n]) for n in range(0, 70) for s in (11, 1234)] + [b"", b"N" * 40, b"ACGTN" * 9]
    rb, ro = pack(ragged)
    rexp, rff, rsc = cpu.pml_batch(rb, ro, threads=2)
    gpu.set_option("out_ring", 0)
    bins0 = gpu.classify_packed(bases, offs, 40, 4)
    lbins0 = gpu.classify_packed(lb, lo, 150, 8)
    gpu.set_option("out_ring", 1)
    for ahead, pair, variant in ((1, 0, -1), (0, 0, -1), (1, 1, -1), (0, 1, -1), (1, 0, 14)):
        gpu.set_option("ahead_rows", ahead)
        gpu.set_option("pair_loads", pair)
        gpu.set_option("pml_variant", variant)
        out, st = gpu.query_pml_packed(bases, offs)
        li = gpu.last_launch()
        ringed = True
        assert li["ahead"] == ahead and li["staged"] == (CAP if ahead == 0 else CAP_AHEAD) - (64 if ringed else 0), li   # the ring takes 4 KB of the block's LDS
        assert li["kernel"].endswith(", %d, 1>" % pair) == ringed, li
        assert (out == exp).all(), (mode, ahead, pair, variant)
        assert (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0), (mode, ahead, pair, variant)
        rout, rst = gpu.query_pml_packed(rb, ro)
        assert (rout == rexp).all() and (rst.fast_forwards, rst.scans) == (rff, rsc), (mode, ahead, pair, variant)
        if variant == -1:
            bins = gpu.classify_packed(bases, offs, 40, 4)
            assert all((x == y).all() for x, y in zip(bins, bins0)), (ahead, pair)
    gpu.set_option("pml_variant", -1)
    gpu.set_option("pair_loads", -1)
    gpu.set_option("ahead_rows", 1)
    # long reads: the policy picks the ring by itself -- one lane per read, and cut into segments
    gpu.set_option("out_ring", -1)
    for seg_len in (0, 512):
        gpu.set_option("seg_len", seg_len)
        gpu.set_option("seg_probe", 0)
        lout, lst = gpu.query_pml_packed(lb, lo)
        assert gpu.last_launch()["segmented"] == (1 if seg_len else 0) and gpu.last_launch()["kernel"].endswith(", 1, 0, 1>")
        assert (lout == lexp).all() and (lst.fast_forwards, lst.scans, lst.errors) == (lff, lsc, 0), seg_len
    gpu.set_option("seg_len", 0)
    lbins = gpu.classify_packed(lb, lo, 150, 8)
    assert all((x == y).all() for x, y in zip(lbins, lbins0))
    # ... and not for short ones: same staging as ever
    out, st = gpu.query_pml_packed(bases, offs)
    assert gpu.last_launch()["staged"] == CAP_AHEAD and (out == exp).all()
    gpu.close()
    cpu.close()


@pytest.mark.parametrize("mode", [6, 3, 8])
def test_zml_pair_shared_gathers_vs_oracle(built_lib, golden_image, mode):
    """"pair_loads" 1 on the ZML parse (zml_kernel_flat<..., PSH = 1>): the two windows of an iteration are fetched by pairs of
    lanes (what the policy does by itself for tables of 2 GB and more).  Same ZML vectors, error bytes and counters as the
    oracle and as the lane-private loads: odd numbers of reads, reads that end while their partner parses on, both widths."""
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = _ref()
    img = golden_image(mode) if mode != 3 else B.build_index_from_seqs([ref], 3)     # (3: a threshold-less `regular` index)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    rng = np.random.default_rng(9970 + mode)
    reads = mutated_reads(rng, ref, 2501, 1, 400) + [bytes(ref[s: s + L]) for s, L in ((100, 1), (200, 8), (300, 17), (5000, 150), (9000, 4000))]
    reads += [b"", b"A", b"N", b"ACGT" * 40, bytes(ref[-300:]), bytes(ref[:300])]
    bases, offs = pack(reads)
    exp = cpu.zml_batch(bases, offs, threads=8)
    gpu.set_option("seg_len", 0)
    gpu.set_option("zml_variant", 1)
    kmode = 3 if mode == 3 else 6
    for idx64 in (0, 1):
        gpu.set_option("idx64", idx64)
        gpu.set_option("pair_loads", 0)
        out0, st0 = gpu.query_zml_packed(bases, offs)
        assert gpu.last_launch()["kernel"].endswith(", 0>") and (out0 == exp).all() and st0.errors == 0
        gpu.set_option("pair_loads", 1)
        out, st = gpu.query_zml_packed(bases, offs)
        li = gpu.last_launch()
        assert li["kernel"].startswith("zml_kernel_flat<%d, " % kmode) and li["kernel"].endswith(", 0, 0, 1, 0>") and li["idx64"] == idx64, li
        assert (out == exp).all(), (mode, idx64)
        assert (st.fast_forwards, st.scans, st.errors) == (st0.fast_forwards, st0.scans, 0), (mode, idx64)
    gpu.set_option("idx64", 0)
    bb, bo = _big_batch(ref, rng, max_len=200, n_long=20)
    bexp = cpu.zml_batch(bb, bo, threads=8)
    bout, bst = gpu.query_zml_packed(bb, bo)
    assert gpu.last_launch()["kernel"].endswith(", 0, 0, 1, 0>") and (bout == bexp).all() and bst.errors == 0
    gpu.close()
    cpu.close()


@pytest.mark.parametrize("mode", [6, 8, 7])
def test_reposition_hints_vs_oracle(built_lib, golden_image, mode):
    """Round 5 -- reposition hints in the look-ahead rows' spare bits (DevIndex::hints, "repo_hints"): a mismatch whose scan leaves
    the row window jumps to where the scan ends (reposition_up / _down, src/move_structure_query.cpp:188-232) when that is within
    7 rows of the window's edge.  Same PML vectors, error bytes and fast-forward / scan / reposition counters as the oracle and
    as the same launch without the hints -- on noisy reads (every few bases a mismatch: many scans that leave their window),
    on poly-base reads (scans that run far, past the hints' reach, and into the table's ends), with both row-index widths, the
    pair-shared gathers, in-window repositions off (every reposition then depends on the hints declining in-window targets),
    segments -- and fewer lane iterations."""
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = _ref()
    img = golden_image(mode) if mode != 7 else B.build_index_from_seqs([ref], 7)
    gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
    rng = np.random.default_rng(5150 + mode)
    refa = np.frombuffer(ref, np.uint8)
    reads = []
    for sub in (0.08, 0.2, 0.5):                                 # noisy to nearly random
        for _ in range(400):
            L = int(rng.integers(1, 700))
            s = int(rng.integers(0, len(ref) - L))
            r = refa[s:s + L].copy()
            m = rng.random(L) < sub
            r[m] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(m.sum()))]
            reads.append(r.tobytes())
    reads += [b"A" * 300, b"C" * 300, b"G" * 300, b"T" * 300, b"ACGT" * 100, b"TTTTTTTTGGGGGGGGCCCCCCCCAAAAAAAA" * 12, b"TG" * 200]
    reads += [bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 400)) for _ in range(200)]      # random reads: a reposition per base
    bases, offs = pack(reads)
    exp, ff, sc = cpu.pml_batch(bases, offs, threads=8)
    gpu.set_option("ahead_rows", 1)
    steps = {}
    for hints in (1, 0):
        gpu.set_option("repo_hints", hints)
        for idx64, pair, inwin in ((0, 0, 1), (1, 0, 1), (0, 1, 1), (0, 0, 0), (1, 1, 0)):
            gpu.set_option("idx64", idx64)
            gpu.set_option("pair_loads", pair)
            gpu.set_option("inwin_repo", inwin)
            out, st, err, rc = gpu.query_pml_packed(bases, offs, want_err=True)
            li = gpu.last_launch()
            assert li["ahead"] == 1 and li["idx64"] == idx64, li
            assert rc == 0 and not err.any()
            assert (out == exp).all(), (mode, hints, idx64, pair, inwin)
            assert (st.fast_forwards, st.scans, st.errors) == (ff, sc, 0), (mode, hints, idx64, pair, inwin)
            steps[(hints, idx64, pair, inwin)] = st.lane_steps
    gpu.set_option("idx64", 0)
    gpu.set_option("pair_loads", -1)
    gpu.set_option("inwin_repo", 1)
    # the jumps are taken: fewer lane iterations with the hints than without, in every configuration
    for key in ((0, 0, 1), (1, 0, 1), (0, 1, 1), (0, 0, 0), (1, 1, 0)):
        assert steps[(1,) + key] < 0.97 * steps[(0,) + key], (key, steps)
    # segments (K1 / K3 walk on the hinted rows too; the stitch kernels on the plain ones)
    long_reads = []
    for _ in range(24):
        s = int(rng.integers(0, len(ref) - 9000))
        r = refa[s:s + 9000].copy()
        m = rng.random(9000) < 0.08
        r[m] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(m.sum()))]
        long_reads.append(r.tobytes())
    lb, lo = pack(long_reads)
    lexp, lff, lsc = cpu.pml_batch(lb, lo, threads=4)
    gpu.set_option("seg_probe", 0)
    for hints in (1, 0):
        gpu.set_option("repo_hints", hints)
        lout, lst = gpu.query_pml_packed(lb, lo)
        assert gpu.last_launch()["segmented"] == 1 and lst.segments > len(long_reads)
        assert (lout == lexp).all() and (lst.fast_forwards, lst.scans, lst.errors) == (lff, lsc, 0), hints
    gpu.set_option("seg_probe", 1)
    gpu.set_option("repo_hints", 1)
    # count and ZML read the same copy: its hint bits are not theirs
    creads = [bytes(ref[s:s + 120]) for s in range(10, 40010, 400)] + reads[:200]
    gpu.set_option("count_variant", 0)
    assert gpu.query_count(creads) == [cpu.count(r) for r in creads]
    assert gpu.last_launch()["kernel"] == "count_kernel_v0<6, 1>"
    gpu.set_option("zml_ahead", 1)
    for r, g in zip(creads[:150], gpu.query_zml(creads[:150])):
        assert (g == cpu.zml(r)).all()
    assert gpu.last_launch()["ahead"] == 1
    gpu.close()
    cpu.close()
