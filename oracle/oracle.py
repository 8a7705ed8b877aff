"""ctypes front for oracle/libmovi_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; nothing under movi_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmovi_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "movi_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmovi_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.oracle_open.restype = C.c_void_p
        L.oracle_open.argtypes = [C.c_void_p, C.c_size_t]
        L.oracle_close.argtypes = [C.c_void_p]
        for f in ("oracle_r", "oracle_length", "oracle_end_bwt_idx"):
            getattr(L, f).restype = C.c_uint64
            getattr(L, f).argtypes = [C.c_void_p]
        L.oracle_mode.argtypes = [C.c_void_p]
        L.oracle_pml.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_pml_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                       C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_pml_logs.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_count.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.oracle_count_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                         C.c_void_p, C.c_int]
        L.oracle_zml.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.oracle_zml_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
        L.oracle_lf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_get_ids.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_get_ids.restype = None
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


class Oracle:
    """CPU restatement of the reference query path on one index.movi image."""

    def __init__(self, index_bytes):
        # (oracle_open copies what it keeps: a numpy image -- the bench's 8 GB table -- is passed as it is)
        buf = (np.ascontiguousarray(index_bytes) if isinstance(index_bytes, np.ndarray) and index_bytes.dtype == np.uint8
               else np.frombuffer(bytes(index_bytes), np.uint8))
        self._h = lib().oracle_open(buf.ctypes.data, buf.size)
        if not self._h:
            raise OracleError("not a v2 index.movi image of a supported type (modes 2, 3, 5, 6, 7, 8)")
        self.r = lib().oracle_r(self._h)
        self.length = lib().oracle_length(self._h)
        self.end_bwt_idx = lib().oracle_end_bwt_idx(self._h)
        self.mode = lib().oracle_mode(self._h)

    def close(self):
        if self._h:
            lib().oracle_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def pml(self, read, stats=False):
        """u16 PMLs in emission order (last base first), as in the BPF file."""
        a = np.frombuffer(bytes(read), np.uint8)
        out = np.zeros(a.size, np.uint16)
        ff, sc = C.c_uint64(0), C.c_uint64(0)
        rc = lib().oracle_pml(self._h, a.ctypes.data if a.size else None, a.size, out.ctypes.data,
                              C.byref(ff), C.byref(sc))
        if rc:
            raise OracleError("oracle_pml rc=%d" % rc)
        return (out, ff.value, sc.value) if stats else out

    def pml_logs(self, read):
        """(PMLs, per-base fast-forwards, per-base scan rows) as `movi query --logs` collects them, emission order."""
        a = np.frombuffer(bytes(read), np.uint8)
        out, ff, sc = np.zeros(a.size, np.uint16), np.zeros(a.size, np.uint16), np.zeros(a.size, np.uint16)
        rc = lib().oracle_pml_logs(self._h, a.ctypes.data if a.size else None, a.size, out.ctypes.data, ff.ctypes.data,
                                   sc.ctypes.data)
        if rc:
            raise OracleError("oracle_pml_logs rc=%d" % rc)
        return out, ff, sc

    def pml_batch(self, seqs, offs, threads=1, strands=16):
        """seqs: uint8 concatenated bases; offs: uint64[n+1].  Returns (out, ff, scan)."""
        seqs = np.ascontiguousarray(seqs, np.uint8)
        offs = np.ascontiguousarray(offs, np.uint64)
        out = np.zeros(seqs.size, np.uint16)
        ff, sc = C.c_uint64(0), C.c_uint64(0)
        rc = lib().oracle_pml_batch(self._h, seqs.ctypes.data, offs.ctypes.data, offs.size - 1,
                                    out.ctypes.data, threads, strands, C.byref(ff), C.byref(sc))
        if rc:
            raise OracleError("oracle_pml_batch rc=%d" % rc)
        return out, ff.value, sc.value

    def count(self, read):
        """(matched, count): printed by the reference as `matched/len\\tcount`."""
        a = np.frombuffer(bytes(read), np.uint8)
        m, c = C.c_uint64(0), C.c_uint64(0)
        rc = lib().oracle_count(self._h, a.ctypes.data if a.size else None, a.size, C.byref(m), C.byref(c))
        if rc:
            raise OracleError("oracle_count rc=%d" % rc)
        return m.value, c.value

    def count_batch(self, seqs, offs, threads=1):
        seqs = np.ascontiguousarray(seqs, np.uint8)
        offs = np.ascontiguousarray(offs, np.uint64)
        n = offs.size - 1
        m = np.zeros(n, np.uint64)
        c = np.zeros(n, np.uint64)
        rc = lib().oracle_count_batch(self._h, seqs.ctypes.data, offs.ctypes.data, n,
                                      m.ctypes.data, c.ctypes.data, threads)
        if rc:
            raise OracleError("oracle_count_batch rc=%d" % rc)
        return m, c

    def zml(self, read):
        """u16 Ziv-Merhav match lengths in emission order (last base first)."""
        a = np.frombuffer(bytes(read), np.uint8)
        out = np.zeros(a.size, np.uint16)
        rc = lib().oracle_zml(self._h, a.ctypes.data if a.size else None, a.size, out.ctypes.data)
        if rc:
            raise OracleError("oracle_zml rc=%d" % rc)
        return out

    def zml_batch(self, seqs, offs, threads=1):
        seqs = np.ascontiguousarray(seqs, np.uint8)
        offs = np.ascontiguousarray(offs, np.uint64)
        out = np.zeros(seqs.size, np.uint16)
        rc = lib().oracle_zml_batch(self._h, seqs.ctypes.data, offs.ctypes.data, offs.size - 1,
                                    out.ctypes.data, threads)
        if rc:
            raise OracleError("oracle_zml_batch rc=%d" % rc)
        return out

    def get_ids(self):
        """MoveStructure::get_id of every row; r where the reference throws."""
        ids = np.zeros(self.r, np.uint64)
        lib().oracle_get_ids(self._h, ids.ctypes.data)
        return ids

    def lf(self, idx, offset):
        i, o = C.c_uint64(idx), C.c_uint64(offset)
        rc = lib().oracle_lf(self._h, C.byref(i), C.byref(o))
        if rc:
            raise OracleError("oracle_lf rc=%d" % rc)
        return i.value, o.value
