"""ctypes declarations for every symbol of include/movi_hip.h."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


class MoviError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("movi_hip error %d: %s" % (code, msg))
        self.code = code


class IndexDescC(C.Structure):
    _fields_ = [
        ("mode", C.c_uint32),
        ("alphabet_size", C.c_uint32),
        ("r", C.c_uint64),
        ("length", C.c_uint64),
        ("end_bwt_idx", C.c_uint64),
        ("end_bwt_idx_thresholds", C.c_uint64 * 4),
        ("alphabet", C.c_uint8 * 8),
        ("code_of", C.c_uint8 * 256),
        ("first_runs", C.c_uint64 * 8),
        ("first_offsets", C.c_uint64 * 8),
        ("last_runs", C.c_uint64 * 8),
        ("last_offsets", C.c_uint64 * 8),
        ("n_blocks", C.c_uint64),
        ("block_size", C.c_uint64),
        ("id_blocks", C.c_void_p),
        ("n_separator_thresholds", C.c_uint64),
        ("separator_thresholds", C.c_void_p),
        ("n_separator_map", C.c_uint64),
        ("separator_map", C.c_void_p),
        ("tally_checkpoints", C.c_uint32),
        ("reserved_", C.c_uint32),
        ("n_tally", C.c_uint64),
        ("tally_ids", C.c_void_p),
    ]


class QueryStatsC(C.Structure):
    _fields_ = [
        ("bases", C.c_uint64),
        ("fast_forwards", C.c_uint64),
        ("scans", C.c_uint64),
        ("repositions", C.c_uint64),
        ("errors", C.c_uint64),
        ("lane_steps", C.c_uint64),
        ("wave_steps", C.c_uint64),
        ("segments", C.c_uint64),
        ("rewalked", C.c_uint64),
    ]


class LaunchInfoC(C.Structure):
    _fields_ = [
        ("kernel", C.c_char * 96),
        ("variant", C.c_int32),
        ("block_threads", C.c_int32),
        ("waves_per_cu", C.c_int32),
        ("segmented", C.c_int32),
        ("idx64", C.c_int32),
        ("staged", C.c_int32),
        ("ahead", C.c_int32),
        ("reserved_", C.c_int32),
    ]


# name -> (restype, argtypes); mirrors include/movi_hip.h one to one.
SYMBOLS = {
    "movi_last_error": (C.c_char_p, []),
    "movi_version": (C.c_int, []),
    "movi_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "movi_index_load": (C.c_int, [C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]),
    "movi_index_parse": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(IndexDescC),
                                   C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "movi_index_create": (C.c_int, [C.c_int, C.POINTER(IndexDescC), C.c_void_p, C.POINTER(C.c_void_p)]),
    "movi_index_create_from_device_rows": (C.c_int, [C.c_int, C.POINTER(IndexDescC), C.c_void_p,
                                                     C.POINTER(C.c_void_p)]),
    "movi_index_replicate": (C.c_int, [C.POINTER(IndexDescC), C.c_void_p, C.POINTER(C.c_int), C.c_int,
                                       C.POINTER(C.c_void_p)]),
    "movi_index_load_replicated": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "movi_index_prepare": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64)]),
    "movi_index_destroy": (C.c_int, [C.c_void_p]),
    "movi_index_get_desc": (C.c_int, [C.c_void_p, C.POINTER(IndexDescC)]),
    "movi_index_device_rows": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "movi_pml_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "movi_pml_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                C.POINTER(QueryStatsC)]),
    "movi_pml_mask_words": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "movi_pml_mask_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "movi_pml_expand_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p,
                                         C.c_void_p]),
    "movi_pml_mask_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                     C.POINTER(QueryStatsC)]),
    "movi_pml_expand_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]),
    "movi_pml_logs_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.POINTER(QueryStatsC)]),
    "movi_zml_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "movi_zml_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                C.POINTER(QueryStatsC)]),
    "movi_last_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(QueryStatsC)]),
    "movi_last_launch": (C.c_int, [C.c_void_p, C.POINTER(LaunchInfoC)]),
    "movi_launch_log": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "movi_index_info": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]),
    "movi_count_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "movi_count_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.POINTER(QueryStatsC)]),
    "movi_classify_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "movi_pml_classify_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32,
                                           C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    "movi_pml_classify_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(QueryStatsC)]),
    "movi_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "movi_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "movi_host_free": (C.c_int, [C.c_void_p]),
    "movi_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "movi_host_unregister": (C.c_int, [C.c_void_p]),
}


def lib_path():
    # MOVI_HIP_LIB: another build of the same C-ABI (A/B measurements against an earlier round's library)
    return os.environ.get("MOVI_HIP_LIB") or os.path.join(_HERE, "lib", "libmovi_hip.so")


_lib = None


def lib():
    """The loaded C-ABI library.  Raises (never falls back) when it is not built."""
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise MoviError(-100, "%s is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                  "or `make -C movi_amd/csrc`" % p)
        L = C.CDLL(p)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)          # AttributeError if the header and the .so drift
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise MoviError(rc, (lib().movi_last_error() or b"").decode("utf-8", "replace"))
