"""CPU suite: the host side of the reset-mask path (include/movi_hip.h "PML as reset masks") -- movi_pml_expand_host is pure host
code, so it is held here to the oracle's vectors on the golden index: masks = (PML == 0) bits in the header's layout, expanded by
the vector code, the plain loop's answer, any thread count, offsets that do not start at 0, the u16 clamp."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def test_mask_words_rule(built_lib):
    """Every read starts a word of its own, ranges never overlap, and the batch fits movi_pml_mask_words -- for any first_base."""
    from movi_amd import engine as E
    rng = np.random.default_rng(1)
    for trial in range(200):
        n = int(rng.integers(1, 40))
        lens = rng.integers(0, 200, n)
        offs = np.concatenate(([0], np.cumsum(lens)))
        fb = int(rng.integers(0, 1000))
        total = E.mask_words(n, int(offs[-1]), fb)
        end_prev = 0
        for i in range(n):
            w0 = ((fb + int(offs[i])) >> 5) - (fb >> 5) + i
            assert w0 >= end_prev
            end_prev = w0 + (int(lens[i]) + 31) // 32
        assert end_prev <= ((int(offs[-1]) + (fb & 31)) >> 5) + n <= total


@pytest.mark.parametrize("mode", [6, 8])
def test_expand_host_equals_oracle_vectors(built_lib, golden_image, mode):
    from movi_amd import engine as E
    from oracle import build_index as B
    from oracle.oracle import Oracle
    cpu = Oracle(golden_image(mode))
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(60 + mode)
    reads = []
    for _ in range(300):
        L = int(rng.integers(1, 700))
        s = int(rng.integers(0, len(ref) - L))
        r = bytearray(ref[s:s + L])
        for k in np.flatnonzero(rng.random(L) < 0.03):
            r[k] = b"ACGTN"[int(rng.integers(0, 5))]
        reads.append(bytes(r))
    reads += [b"", b"A", b"N" * 40, ref[100:100 + 3000], b"", b"acgt"]
    bases = np.frombuffer(b"".join(reads), np.uint8)
    offs = np.concatenate(([0], np.cumsum([len(r) for r in reads]))).astype(np.uint64)
    exp, _, _ = cpu.pml_batch(bases, offs, threads=4)
    words, valid = E.masks_of_pml(exp, offs)
    words[~valid] = 0xDEADBEEF                                  # gap words are unspecified: nobody may read them
    for th in (0, 1, 2, 5):
        assert (E.expand_masks_host(words, offs, threads=th) == exp).all(), th
    # offsets that do not start at 0 (a window into a larger array): masks relative to offsets[0], vector at the absolute positions
    shift = 12345
    out = np.zeros(shift + bases.size, np.uint16)
    E.expand_masks_host(words, offs + np.uint64(shift), threads=2, out=out)
    assert (out[shift:] == exp).all() and not out[:shift].any()


def test_expand_host_u16_clamp_and_long_runs(built_lib):
    from movi_amd import engine as E
    offs = np.array([0, 70000, 70001, 70001 + 65600], np.uint64)
    pml = np.zeros(int(offs[-1]), np.uint16)
    pml[:70000] = np.minimum(np.arange(1, 70001), 65535)         # one long run of matches: clamps at 65535
    pml[70000] = 0
    run = np.arange(1, 65601)
    run[100:] = np.arange(0, 65500)                               # a reset at step 100, then a run that stops short of the clamp
    pml[70001:] = np.minimum(run, 65535)
    words, _ = E.masks_of_pml(pml, offs)
    for th in (1, 4):
        assert (E.expand_masks_host(words, offs, threads=th) == pml).all()


def test_expand_host_rejects_bad_arguments(built_lib):
    import ctypes as C
    from movi_amd._lib import lib
    offs = np.array([0, 10, 5], np.uint64)
    words = np.zeros(8, np.uint32)
    out = np.zeros(16, np.uint16)
    assert lib().movi_pml_expand_host(words.ctypes.data, offs.ctypes.data, 2, out.ctypes.data, 1) == -1     # offsets decrease
    assert lib().movi_pml_expand_host(None, offs.ctypes.data, 2, out.ctypes.data, 1) == -1
    assert lib().movi_pml_expand_host(None, None, 0, None, 1) == 0
    n = C.c_uint64(0)
    assert lib().movi_pml_mask_words(3, 100, 0, None) == -1 and lib().movi_pml_mask_words(3, 100, 0, C.byref(n)) == 0 and n.value >= 3 + 3
