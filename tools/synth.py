"""synth.py -- ctypes front of tools/synth.c (synthetic indexes + reads).

Bench / test tooling; see synth.c for the recipe.  Imports nothing from oracle/
(this module feeds the measured path).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmovi_synth.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "synth.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-fopenmp", "-shared", "-std=c11", "-o", _SO, src, "-lm"])
    return _SO


def _L():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.synth_create.restype = C.c_void_p
        L.synth_create.argtypes = [C.c_uint64, C.c_int, C.c_uint64, C.c_double]
        L.synth_free.argtypes = [C.c_void_p]
        for f in ("synth_r", "synth_n", "synth_end_bwt_idx", "synth_image_size"):
            getattr(L, f).restype = C.c_uint64
            getattr(L, f).argtypes = [C.c_void_p]
        L.synth_write_image.argtypes = [C.c_void_p, C.c_void_p]
        L.synth_reads.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_double, C.c_double,
                                  C.c_void_p, C.c_int]
        _lib = L
    return _lib


class SynthIndex:
    def __init__(self, h, mode):
        self._h = h
        self.mode = mode
        self.r = _L().synth_r(h)
        self.n = _L().synth_n(h)
        self.end_bwt_idx = _L().synth_end_bwt_idx(h)
        self._img = None

    def image(self):
        """The bytes of a v2 index.movi (numpy uint8 array; cached)."""
        if self._img is None:
            buf = np.empty(_L().synth_image_size(self._h), np.uint8)
            _L().synth_write_image(self._h, buf.ctypes.data)
            self._img = buf
        return self._img

    def __del__(self):
        try:
            if self._h:
                _L().synth_free(self._h)
                self._h = None
        except Exception:
            pass


def synth_index(r, mode=6, seed=20260529, mean_run=40.0):
    h = _L().synth_create(int(r), int(mode), int(seed), float(mean_run))
    if not h:
        raise RuntimeError("synth_create failed (r=%d mode=%d)" % (r, mode))
    return SynthIndex(h, mode)


def synth_reads(ix, n_reads, read_len, seed=1, sub_rate=0.01, n_rate=0.001, lens=None, threads=0):
    """(bases uint8[sum len], offsets uint64[n_reads+1]); `lens` overrides read_len."""
    if lens is None:
        lens = np.full(n_reads, read_len, np.uint64)
    lens = np.asarray(lens, np.uint64)
    offs = np.zeros(n_reads + 1, np.uint64)
    np.cumsum(lens, out=offs[1:])
    bases = np.empty(int(offs[-1]), np.uint8)
    _L().synth_reads(ix._h, n_reads, offs.ctypes.data, int(seed), float(sub_rate), float(n_rate),
                     bases.ctypes.data, int(threads))
    return bases, offs
