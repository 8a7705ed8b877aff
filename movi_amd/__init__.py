"""movi_amd -- MI355X (gfx950) engine for Movi's PML / count query path.

The compute lives in ``movi_amd/lib/libmovi_hip.so`` (hand-written HIP kernels
behind the C-ABI of ``include/movi_hip.h``); ``movi_amd/bin/movi`` is the C++
host CLI (`movi query`, `movi view`).  This Python package is a thin ctypes
binding used by the tests and by bench.py -- it contains no compute and no CPU
fallback: without the built library every entry point raises.
"""
from ._lib import MoviError, lib, lib_path  # noqa: F401
from .engine import MoveIndex, IndexDesc, parse_index_image, pinned_empty  # noqa: F401

__all__ = ["MoviError", "MoveIndex", "IndexDesc", "parse_index_image", "pinned_empty", "lib", "lib_path"]
