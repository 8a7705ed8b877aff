"""CPU suite: the C-ABI library builds, loads, exports every symbol that
include/movi_hip.h declares, and its host-only entry points behave.  No compute
call is made here (there is no GPU in the build container)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from oracle.oracle import Oracle


def header_functions():
    text = open(os.path.join(ROOT, "include", "movi_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(movi_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built_lib):
    from movi_amd._lib import SYMBOLS
    L = C.CDLL(built_lib)
    declared = header_functions()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(L, name), "libmovi_hip.so does not export %s" % name
    # the ctypes table covers the whole header, nothing more
    assert sorted(SYMBOLS) == declared


def test_parse_golden_index_matches_oracle(built_lib, golden_image):
    import movi_amd
    for mode in (6, 8):
        img = golden_image(mode)
        desc, _, off, nbytes = movi_amd.parse_index_image(img)
        o = Oracle(img)
        assert (desc.mode, desc.r, desc.length, desc.end_bwt_idx) == (mode, o.r, o.length, o.end_bwt_idx)
        assert desc.r == 118209 and off == 2215
        assert nbytes == desc.r * (8 if mode == 6 else 6)
        assert desc.alphabet == b"ACGT"
        code = np.frombuffer(desc.code_of, np.uint8)
        assert [int(code[c]) for c in b"ACGT"] == [0, 1, 2, 3]
        assert (np.delete(code, list(b"ACGT")) == 0xFF).all()
        if mode == 8:
            assert desc.n_blocks == 1 and desc.block_size == 1 << 20


def test_parse_rejects_bad_images(built_lib, golden_image):
    import movi_amd
    img = bytearray(golden_image(6))
    with pytest.raises(movi_amd.MoviError) as e:
        movi_amd.parse_index_image(bytes(img[:40]))
    assert e.value.code == -2
    bad = bytearray(img); bad[0] ^= 0xFF
    with pytest.raises(movi_amd.MoviError):
        movi_amd.parse_index_image(bytes(bad))
    for legacy in (0, 1, 4):                                  # large / constant / split: Movi-1 style rows, out of scope
        other_mode = bytearray(img); other_mode[7] = legacy
        with pytest.raises(movi_amd.MoviError) as e:
            movi_amd.parse_index_image(bytes(other_mode))
        assert "not supported" in str(e.value)
    with pytest.raises(movi_amd.MoviError):
        movi_amd.parse_index_image(bytes(img[: len(img) // 2]))
    # u64 wrap-around: r + 2^61 makes r * 8 wrap to the true table size, so a naive `pos + r * 8 <= n` accepts it
    import struct
    for mode, row_b in ((6, 8), (8, 6)):
        gi = bytearray(golden_image(mode))
        (r,) = struct.unpack_from("<Q", gi, 24)
        for bogus in (r + (1 << 61), r + (1 << 63), (1 << 64) - 1, 1 << 36):
            w = bytearray(gi)
            struct.pack_into("<Q", w, 24, bogus)
            with pytest.raises(movi_amd.MoviError) as e:
                movi_amd.parse_index_image(bytes(w))
            assert e.value.code == -2, (mode, bogus)
    # blocked: n_blocks / block_size whose product wraps must not pass the coverage check
    gi = bytearray(golden_image(8))
    desc, c, off, nbytes = movi_amd.parse_index_image(bytes(gi))
    at = len(gi) - 8                                          # trailing u64 block_size (move_structure_io.cpp:321-323)
    assert struct.unpack_from("<Q", gi, at)[0] == 1 << 20
    for bogus in (0, 1, 1 << 10):                             # 1 block of 1 / 1024 rows does not cover 118209 rows
        w = bytearray(gi)
        struct.pack_into("<Q", w, at, bogus)
        with pytest.raises(movi_amd.MoviError):
            movi_amd.parse_index_image(bytes(w))


def test_no_cpu_fallback(built_lib, golden_image):
    """Without a GPU the product path must fail loudly, never compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import movi_amd
    with pytest.raises(movi_amd.MoviError) as e:
        movi_amd.MoveIndex.from_image(golden_image(6))
    assert e.value.code in (-4, -5)


def test_host_memory_entry_points_without_gpu(built_lib):
    """movi_host_alloc / movi_host_register fail with a status and a message (no device here), never crash; NULL frees are no-ops."""
    import ctypes as C
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from movi_amd._lib import lib
    p = C.c_void_p()
    assert lib().movi_host_alloc(1024, C.byref(p)) in (-4, -5) and not p.value and lib().movi_last_error()
    assert lib().movi_host_alloc(1024, None) == -1
    a = np.zeros(4096, np.uint8)
    assert lib().movi_host_register(a.ctypes.data, a.nbytes) in (-4, -5)
    assert lib().movi_host_register(None, 16) == -1 and lib().movi_host_register(a.ctypes.data, 0) == -1
    assert lib().movi_host_free(None) == 0 and lib().movi_host_unregister(None) == 0


def test_product_does_not_touch_oracle():
    """Nothing under movi_amd/ or tools/synth.py may reference oracle/."""
    for base, _, files in os.walk(os.path.join(ROOT, "movi_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                assert "oracle" not in open(os.path.join(base, f), errors="replace").read().lower(), f
    for f in ("synth.py", "synth.c"):            # the generator feeds the measured path too
        code = open(os.path.join(ROOT, "tools", f)).read()
        assert not re.search(r"^\s*(from|import)\s+oracle|#include.*oracle|libmovi_oracle", code, re.M), f


def test_parse_separators_index(built_lib):
    """A `movi build --separators` image ('%' + ACGT, reference KAT size 948232 B, tests/test_build.cpp:79):
    parsed, '%' illegal in reads, the two separator tables located; other 5-symbol alphabets rejected."""
    import movi_amd
    from conftest import GOLDEN
    from oracle import build_index as B
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    for mode, size in ((6, 948232), (8, 711854)):
        img = B.build_index_from_seqs([ref], mode, separators=True)
        assert len(img) == size
        desc, c, off, nbytes = movi_amd.parse_index_image(img)
        assert desc.alphabet == b"%ACGT" and desc.alphabet_size == 5 and desc.r == 118207
        code = np.frombuffer(desc.code_of, np.uint8)
        assert [int(code[x]) for x in b"ACGT"] == [1, 2, 3, 4] and code[ord("%")] == 0xFF
        assert (int(c.n_separator_thresholds), int(c.n_separator_map)) == (3, 3)
        assert len(desc.first_runs) == 6
        o = Oracle(img)
        assert (o.r, o.end_bwt_idx) == (desc.r, desc.end_bwt_idx)
        with pytest.raises(movi_amd.MoviError):
            movi_amd.parse_index_image(img[:-8])                     # truncated separator map
    bad = bytearray(B.build_index_from_seqs([ref], 6, separators=True))
    at = bad.index(b"%ACGT")
    bad[at] = ord("#")
    with pytest.raises(movi_amd.MoviError) as e:
        movi_amd.parse_index_image(bytes(bad))
    assert "separator" in str(e.value)


def test_parse_sampled_thresholds_index(built_lib):
    """Mode 7 (reference KAT 475326 B, tests/test_build.cpp:45-47): 3-byte rows, the tally table located."""
    import movi_amd
    from conftest import GOLDEN
    from oracle import build_index as B
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    img = B.build_index_from_seqs([ref], 7)
    assert len(img) == 475326
    desc, c, off, nbytes = movi_amd.parse_index_image(img)
    assert (desc.mode, desc.r, desc.row_bytes, nbytes) == (7, 118209, 3, 118209 * 3)
    assert int(c.tally_checkpoints) == 20 and int(c.n_tally) == 118209 // 20 + 2
    with pytest.raises(movi_amd.MoviError):
        movi_amd.parse_index_image(img[: off + nbytes + 100])           # truncated tally table


def test_parse_sampled_index(built_lib):
    """Mode 5 (reference KAT 437006 B, tests/test_build.cpp:41-43): accepted; with separators there is no threshold section."""
    import movi_amd
    from conftest import GOLDEN
    from oracle import build_index as B
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    for sep, size in ((False, 437006), (True, 464203)):
        img = B.build_index_from_seqs([ref], 5, separators=sep)
        assert len(img) == size
        desc, c, off, nbytes = movi_amd.parse_index_image(img)
        assert (desc.mode, desc.row_bytes, nbytes) == (5, 3, desc.r * 3)
        assert int(c.tally_checkpoints) == 20 and int(c.n_separator_thresholds) == 0


def test_parse_regular_and_blocked_indexes(built_lib):
    """Modes 3 / 2 (reference KATs 871479 / 654253 B, tests/test_build.cpp:33,49): accepted; with separators there is no
    threshold section (USE_THRESHOLDS is off); blocked: id blocks located, default block size 2^22."""
    import movi_amd
    from conftest import GOLDEN
    from oracle import build_index as B
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    for mode, sep, size in ((3, False, 871479), (3, True, 871496), (2, False, 654253), (2, True, 654280)):
        img = B.build_index_from_seqs([ref], mode, separators=sep)
        assert len(img) == size
        desc, c, off, nbytes = movi_amd.parse_index_image(img)
        assert (desc.mode, desc.row_bytes, nbytes) == (mode, 8 if mode == 3 else 6, desc.r * (8 if mode == 3 else 6))
        assert int(c.n_separator_thresholds) == 0 and desc.alphabet == (b"%ACGT" if sep else b"ACGT")
        if mode == 2:
            assert desc.n_blocks == 1 and desc.block_size == 1 << 22
            with pytest.raises(movi_amd.MoviError):
                movi_amd.parse_index_image(img[:-12])                   # truncated id blocks


def test_parse_fuzz_sanitized(tmp_path):
    """movi_index_parse under AddressSanitizer + UBSan (CPU build of the ABI's host code): truncations and field smashing of
    one small index per type; no sanitizer report, and every accepted image is self-consistent (tests/fuzz/fuzz_parse.cpp)."""
    import subprocess
    script = os.path.join(ROOT, "tests", "fuzz", "fuzz_parse.sh")
    if not os.path.exists(script):
        pytest.skip("tests/fuzz/fuzz_parse.sh is CPU-container tooling (.gpurunignore keeps it off the GPU box)")
    r = subprocess.run([script, "1500", str(tmp_path)], capture_output=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert b"fuzz ok" in r.stdout and b"no sanitizer report" in r.stdout
