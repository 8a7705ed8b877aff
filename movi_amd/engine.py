"""Host-side mirror of the reference's query interface over the C-ABI.

Reference seam (file:line under /root/reference) -> here:
  MoveStructure::deserialize          src/move_structure_io.cpp:471-511  -> MoveIndex.load / from_image
  MoveStructure::query_pml(MoveQuery&) src/move_structure_query.cpp:234   -> MoveIndex.query_pml
  ReadProcessor::process_latency_hiding src/read_processor.cpp:641        -> MoveIndex.query_pml (batched)
  MoveStructure::query_backward_search src/move_structure_search.cpp:340  -> MoveIndex.query_count

All compute happens in libmovi_hip.so on the GPU; this file only marshals
buffers.  Errors are MoviError (the reference throws std::runtime_error).
"""
import ctypes as C

import numpy as np

from ._lib import IndexDescC, LaunchInfoC, QueryStatsC, check, lib


class IndexDesc:
    """Python view of movi_index_desc_t."""

    def __init__(self, c):
        self.mode = int(c.mode)
        self.alphabet_size = int(c.alphabet_size)
        self.r = int(c.r)
        self.length = int(c.length)
        self.end_bwt_idx = int(c.end_bwt_idx)
        self.end_bwt_idx_thresholds = list(c.end_bwt_idx_thresholds)
        self.alphabet = bytes(c.alphabet[: self.alphabet_size])
        self.code_of = bytes(c.code_of)
        k = self.alphabet_size + 1
        self.first_runs = list(c.first_runs[:k])
        self.first_offsets = list(c.first_offsets[:k])
        self.last_runs = list(c.last_runs[:k])
        self.last_offsets = list(c.last_offsets[:k])
        self.n_blocks = int(c.n_blocks)
        self.block_size = int(c.block_size)
        self.row_bytes = {6: 8, 8: 6, 7: 3, 5: 3, 3: 8, 2: 6}[self.mode]


def parse_index_image(image):
    """movi_index_parse: (IndexDesc, raw C desc, rows_offset, rows_bytes).  Host only, no GPU."""
    buf = np.frombuffer(image, np.uint8)
    c = IndexDescC()
    off, nb = C.c_size_t(0), C.c_size_t(0)
    check(lib().movi_index_parse(buf.ctypes.data, buf.size, C.byref(c), C.byref(off), C.byref(nb)))
    return IndexDesc(c), c, off.value, nb.value


def _pack_reads(reads):
    """list of bytes -> (uint8 bases, uint64 offsets[n+1])."""
    lens = np.fromiter((len(r) for r in reads), np.uint64, len(reads))
    offs = np.zeros(len(reads) + 1, np.uint64)
    np.cumsum(lens, out=offs[1:])
    bases = np.frombuffer(b"".join(bytes(r) for r in reads), np.uint8) if len(reads) else np.zeros(0, np.uint8)
    return np.ascontiguousarray(bases), offs


class _PinnedBlock:
    """Page-locked host bytes (movi_host_alloc); numpy views keep it alive through __array_interface__."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        check(lib().movi_host_alloc(nbytes, C.byref(p)))
        self.ptr, self.nbytes = p.value, nbytes
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 3}

    def __del__(self):
        try:
            if self.ptr:
                lib().movi_host_free(C.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


def pinned_empty(n, dtype):
    """A numpy array of n elements in page-locked host memory (movi_host_alloc): with the reads and the result
    vector in such arrays the *_host entry points overlap upload, walk and download (include/movi_hip.h)."""
    dt = np.dtype(dtype)
    blk = _PinnedBlock(max(int(n) * dt.itemsize, 1))
    return np.asarray(blk)[: int(n) * dt.itemsize].view(dt)


def mask_words(n_reads, n_bases, first_base=0):
    """movi_pml_mask_words: 32-bit words the reset masks of a batch take."""
    n = C.c_uint64(0)
    check(lib().movi_pml_mask_words(int(n_reads), int(n_bases), int(first_base), C.byref(n)))
    return int(n.value)


def expand_masks_host(words, offs, threads=0, out=None):
    """movi_pml_expand_host: reset-mask words -> the u16 PML vector (pure host code, worker threads)."""
    words = np.ascontiguousarray(words, np.uint32)
    offs = np.ascontiguousarray(offs, np.uint64)
    if out is None:
        out = np.zeros(int(offs[-1]), np.uint16)
    check(lib().movi_pml_expand_host(words.ctypes.data, offs.ctypes.data, offs.size - 1, out.ctypes.data, int(threads)))
    return out


def masks_of_pml(pml, offs, first_base=0):
    """The reset-mask words of a PML vector (numpy; test helper): bit = (PML == 0), layout of include/movi_hip.h.
    Returns (words, valid) -- valid marks the words that belong to a read (gap words are unspecified)."""
    offs = np.asarray(offs, np.uint64).astype(np.int64)
    n = offs.size - 1
    rel = offs - offs[0]
    words = np.zeros(mask_words(n, int(rel[-1]), first_base), np.uint32)
    valid = np.zeros(words.size, bool)
    lens = np.diff(rel)
    if n == 0 or rel[-1] == 0:
        return words, valid
    w0 = ((first_base + rel[:-1]) >> 5) - (first_base >> 5) + np.arange(n)
    read_of = np.repeat(np.arange(n), lens)
    k = np.arange(int(rel[-1])) - np.repeat(rel[:-1], lens)
    z = np.asarray(pml[int(offs[0]):int(offs[-1])]) == 0
    widx = w0[read_of] + (k >> 5)
    np.bitwise_or.at(words, widx[z], (np.uint32(1) << (k[z] & 31).astype(np.uint32)))
    valid[widx] = True
    return words, valid


class QueryStats:
    def __init__(self, c):
        self.bases = int(c.bases)
        self.fast_forwards = int(c.fast_forwards)
        self.scans = int(c.scans)
        self.repositions = int(c.repositions)
        self.errors = int(c.errors)
        self.lane_steps = int(c.lane_steps)
        self.wave_steps = int(c.wave_steps)
        self.segments = int(c.segments)
        self.rewalked = int(c.rewalked)

    def __repr__(self):
        return "QueryStats(bases=%d, ff=%d, scans=%d, repositions=%d, errors=%d)" % (
            self.bases, self.fast_forwards, self.scans, self.repositions, self.errors)


class MoveIndex:
    """A move-structure index resident on one GPU."""

    def __init__(self, handle, keepalive=None):
        self._h = handle
        self._keep = keepalive
        c = IndexDescC()
        check(lib().movi_index_get_desc(self._h, C.byref(c)))
        self.desc = IndexDesc(c)

    # -- construction ---------------------------------------------------------
    @classmethod
    def load(cls, index_dir_or_file, device=0):
        h = C.c_void_p()
        check(lib().movi_index_load(device, str(index_dir_or_file).encode(), C.byref(h)))
        return cls(h)

    @classmethod
    def from_image(cls, image, device=0):
        buf = np.frombuffer(image, np.uint8)
        _, c, off, _ = parse_index_image(buf)
        h = C.c_void_p()
        check(lib().movi_index_create(device, C.byref(c), buf.ctypes.data + off, C.byref(h)))
        return cls(h)

    @classmethod
    def from_device_rows(cls, cdesc, d_rows_ptr, device=0, keepalive=None):
        """Adopt a device-resident row table (e.g. an RCCL-broadcast torch tensor)."""
        h = C.c_void_p()
        check(lib().movi_index_create_from_device_rows(device, C.byref(cdesc), C.c_void_p(d_rows_ptr), C.byref(h)))
        return cls(h, keepalive=keepalive)

    @classmethod
    def load_replicated(cls, index_dir_or_file, devices):
        """movi_index_load_replicated: one handle per device; the rows cross PCIe once (to devices[0]) and reach the
        other GPUs through one RCCL broadcast."""
        n = len(devices)
        devs = (C.c_int * n)(*[int(d) for d in devices])
        hs = (C.c_void_p * n)()
        check(lib().movi_index_load_replicated(str(index_dir_or_file).encode(), devs, n, hs))
        return [cls(C.c_void_p(hs[i])) for i in range(n)]

    @classmethod
    def replicate_image(cls, image, devices):
        """movi_index_replicate from an index.movi image in host memory."""
        buf = np.frombuffer(image, np.uint8)
        _, c, off, _ = parse_index_image(buf)
        n = len(devices)
        devs = (C.c_int * n)(*[int(d) for d in devices])
        hs = (C.c_void_p * n)()
        check(lib().movi_index_replicate(C.byref(c), buf.ctypes.data + off, devs, n, hs))
        return [cls(C.c_void_p(hs[i])) for i in range(n)]

    def close(self):
        if getattr(self, "_h", None):
            lib().movi_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_rows(self):
        p, n = C.c_void_p(), C.c_size_t()
        check(lib().movi_index_device_rows(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def set_option(self, key, value):
        check(lib().movi_set_option(self._h, key.encode(), int(value)))

    # -- host-buffer queries ----------------------------------------------------
    def query_pml_packed(self, bases, offs, want_err=False, out=None):
        """bases uint8[n_bases], offs uint64[n+1] -> (u16 PMLs in emission order, QueryStats[, err]).
        `out`: the caller's result array (e.g. pinned_empty(n_bases, np.uint16) next to pinned bases: overlapped path)."""
        bases = np.ascontiguousarray(bases, np.uint8)
        offs = np.ascontiguousarray(offs, np.uint64)
        n = offs.size - 1
        if out is None:
            out = np.zeros(bases.size, np.uint16)
        assert out.dtype == np.uint16 and out.size == bases.size and out.flags.c_contiguous
        err = np.zeros(max(n, 1), np.uint8)
        st = QueryStatsC()
        rc = lib().movi_pml_host(self._h, bases.ctypes.data, offs.ctypes.data, n, out.ctypes.data,
                                 err.ctypes.data, C.byref(st))
        if want_err:
            return out, QueryStats(st), err[:n], rc
        check(rc)
        return out, QueryStats(st)

    def query_pml_mask_packed(self, bases, offs, want_err=False):
        """movi_pml_mask_host: (u32 reset-mask words laid out as include/movi_hip.h says, QueryStats[, err, rc])."""
        bases = np.ascontiguousarray(bases, np.uint8)
        offs = np.ascontiguousarray(offs, np.uint64)
        n = offs.size - 1
        words = np.zeros(mask_words(n, int(offs[-1] - offs[0])), np.uint32)
        err = np.zeros(max(n, 1), np.uint8)
        st = QueryStatsC()
        rc = lib().movi_pml_mask_host(self._h, bases.ctypes.data, offs.ctypes.data, n, words.ctypes.data, err.ctypes.data,
                                      C.byref(st))
        if want_err:
            return words, QueryStats(st), err[:n], rc
        check(rc)
        return words, QueryStats(st)

    def query_pml_logs_packed(self, bases, offs):
        """movi_pml_logs_host: (PMLs, per-base fast-forwards, per-base scan rows, QueryStats), all in emission order."""
        bases = np.ascontiguousarray(bases, np.uint8)
        offs = np.ascontiguousarray(offs, np.uint64)
        out = np.zeros(bases.size, np.uint16)
        ff = np.zeros(bases.size, np.uint16)
        sc = np.zeros(bases.size, np.uint16)
        st = QueryStatsC()
        check(lib().movi_pml_logs_host(self._h, bases.ctypes.data, offs.ctypes.data, offs.size - 1, out.ctypes.data,
                                       ff.ctypes.data, sc.ctypes.data, None, C.byref(st)))
        return out, ff, sc, QueryStats(st)

    def query_pml(self, reads):
        """MoveStructure::query_pml for each read: list of u16 arrays, last base first
        (MoveQuery::matching_lens order)."""
        bases, offs = _pack_reads(reads)
        out, _ = self.query_pml_packed(bases, offs)
        return [out[int(offs[i]): int(offs[i + 1])] for i in range(len(reads))]

    def query_zml_packed(self, bases, offs, out=None):
        """MoveStructure::query_zml (src/move_structure_query.cpp:690-785) for packed reads:
        (u16 match lengths in emission order, QueryStats)."""
        bases = np.ascontiguousarray(bases, np.uint8)
        offs = np.ascontiguousarray(offs, np.uint64)
        n = offs.size - 1
        if out is None:
            out = np.zeros(bases.size, np.uint16)
        assert out.dtype == np.uint16 and out.size == bases.size and out.flags.c_contiguous
        st = QueryStatsC()
        check(lib().movi_zml_host(self._h, bases.ctypes.data, offs.ctypes.data, n, out.ctypes.data, None,
                                  C.byref(st)))
        return out, QueryStats(st)

    def query_zml(self, reads):
        bases, offs = _pack_reads(reads)
        out, _ = self.query_zml_packed(bases, offs)
        return [out[int(offs[i]): int(offs[i + 1])] for i in range(len(reads))]

    def classify_packed(self, bases, offs, bin_width, max_value_thr):
        """PML + Classifier::classify bins on the device: (bins_above, bins_below, sum_max) per read."""
        bases = np.ascontiguousarray(bases, np.uint8)
        offs = np.ascontiguousarray(offs, np.uint64)
        n = offs.size - 1
        a = np.zeros(max(n, 1), np.uint32)
        b = np.zeros(max(n, 1), np.uint32)
        s = np.zeros(max(n, 1), np.uint64)
        st = QueryStatsC()
        check(lib().movi_pml_classify_host(self._h, bases.ctypes.data, offs.ctypes.data, n, int(bin_width),
                                           int(max_value_thr), a.ctypes.data, b.ctypes.data, s.ctypes.data,
                                           None, C.byref(st)))
        return a[:n], b[:n], s[:n]

    def query_count_packed(self, bases, offs, want_err=False):
        """-> (matched, count, QueryStats[, per-read error bytes, return code])"""
        bases = np.ascontiguousarray(bases, np.uint8)
        offs = np.ascontiguousarray(offs, np.uint64)
        n = offs.size - 1
        m = np.zeros(max(n, 1), np.uint64)
        c = np.zeros(max(n, 1), np.uint64)
        err = np.zeros(max(n, 1), np.uint8)
        st = QueryStatsC()
        rc = lib().movi_count_host(self._h, bases.ctypes.data, offs.ctypes.data, n, m.ctypes.data,
                                   c.ctypes.data, err.ctypes.data if want_err else None, C.byref(st))
        if want_err:
            return m[:n], c[:n], QueryStats(st), err[:n], rc
        check(rc)
        return m[:n], c[:n], QueryStats(st)

    def query_count(self, reads):
        """query_backward_search per read: list of (matched, count) as printed by
        output_counts (src/utils.cpp:248-256: `matched/len\\tcount`)."""
        bases, offs = _pack_reads(reads)
        m, c, _ = self.query_count_packed(bases, offs)
        return [(int(m[i]), int(c[i])) for i in range(len(reads))]

    # -- device-pointer queries (bench / torch interop) ---------------------------
    def pml_device(self, d_bases, d_offs, n_reads, n_bases, d_out, d_err=0, stream=0, d_order=0):
        check(lib().movi_pml_device(self._h, C.c_void_p(d_bases), C.c_void_p(d_offs), n_reads, n_bases,
                                    C.c_void_p(d_out), C.c_void_p(d_err) if d_err else None,
                                    C.c_void_p(d_order) if d_order else None,
                                    C.c_void_p(stream) if stream else None))

    def pml_mask_device(self, d_bases, d_offs, n_reads, n_bases, d_words, first_base=0, d_err=0, stream=0, d_order=0):
        check(lib().movi_pml_mask_device(self._h, C.c_void_p(d_bases), C.c_void_p(d_offs), n_reads, n_bases, int(first_base),
                                         C.c_void_p(d_words), C.c_void_p(d_err) if d_err else None,
                                         C.c_void_p(d_order) if d_order else None,
                                         C.c_void_p(stream) if stream else None))

    def pml_expand_device(self, d_words, d_offs, n_reads, n_bases, d_out, first_base=0, stream=0):
        check(lib().movi_pml_expand_device(self._h, C.c_void_p(d_words), C.c_void_p(d_offs), n_reads, n_bases, int(first_base),
                                           C.c_void_p(d_out), C.c_void_p(stream) if stream else None))

    def pml_classify_device(self, d_bases, d_offs, n_reads, n_bases, bin_width, max_value_thr, d_out, d_above,
                            d_below, d_sum, d_err=0, stream=0, d_order=0):
        """PML walk with Classifier::classify bins fused in; d_out = 0 writes no PML vector."""
        check(lib().movi_pml_classify_device(self._h, C.c_void_p(d_bases), C.c_void_p(d_offs), n_reads, n_bases,
                                             int(bin_width), int(max_value_thr),
                                             C.c_void_p(d_out) if d_out else None, C.c_void_p(d_above),
                                             C.c_void_p(d_below), C.c_void_p(d_sum),
                                             C.c_void_p(d_err) if d_err else None,
                                             C.c_void_p(d_order) if d_order else None,
                                             C.c_void_p(stream) if stream else None))

    def zml_device(self, d_bases, d_offs, n_reads, n_bases, d_out, d_err=0, stream=0, d_order=0):
        check(lib().movi_zml_device(self._h, C.c_void_p(d_bases), C.c_void_p(d_offs), n_reads, n_bases,
                                    C.c_void_p(d_out), C.c_void_p(d_err) if d_err else None,
                                    C.c_void_p(d_order) if d_order else None,
                                    C.c_void_p(stream) if stream else None))

    def count_device(self, d_bases, d_offs, n_reads, n_bases, d_matched, d_count, d_err=0, stream=0, d_order=0):
        check(lib().movi_count_device(self._h, C.c_void_p(d_bases), C.c_void_p(d_offs), n_reads, n_bases,
                                      C.c_void_p(d_matched), C.c_void_p(d_count),
                                      C.c_void_p(d_err) if d_err else None,
                                      C.c_void_p(d_order) if d_order else None,
                                      C.c_void_p(stream) if stream else None))

    PREPARE_PML, PREPARE_COUNT, PREPARE_ZML = 1, 2, 4

    def prepare(self, what=1 | 2 | 4, stream=0):
        """movi_index_prepare: build the handle's derived tables now (not inside the first query); returns their bytes."""
        n = C.c_uint64(0)
        check(lib().movi_index_prepare(self._h, int(what), C.c_void_p(stream) if stream else None, C.byref(n)))
        return int(n.value)

    def last_launch(self):
        """movi_last_launch: dict(kernel=..., variant=..., block_threads=..., waves_per_cu=..., segmented=..., idx64=...)."""
        li = LaunchInfoC()
        check(lib().movi_last_launch(self._h, C.byref(li)))
        return {"kernel": li.kernel.decode(), "variant": int(li.variant), "block_threads": int(li.block_threads),
                "waves_per_cu": int(li.waves_per_cu), "segmented": int(li.segmented), "idx64": int(li.idx64), "staged": int(li.staged), "ahead": int(li.ahead)}

    def info(self, key):
        """movi_index_info: bytes of the derived tables the handle holds, statistics their builders tallied."""
        v = C.c_double()
        check(lib().movi_index_info(self._h, key.encode(), C.byref(v)))
        return v.value

    def last_stats(self, stream=0):
        st = QueryStatsC()
        check(lib().movi_last_stats(self._h, C.c_void_p(stream) if stream else None, C.byref(st)))
        return QueryStats(st)
