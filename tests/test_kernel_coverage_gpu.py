"""GPU suite (-m gpu): every instantiation of the walk kernel the library was BUILT with is reachable through the options of
the C-ABI, and each one is held to the oracle once.

pml_kernel_flatp<6, IdxT, CLS, SEP, SEG, STG, AHD, PSH, RING> (movi_amd/csrc/movi_walk.hpp) has 218 instantiations (round 6: + 16 with
RING = 2, reset masks out: index width x separators x look-ahead rows x pair-shared gathers; + 22 with AHD = 2, the deep rows); the launch
policy picks among them from the table (separators, size), the batch (read lengths, size) and a dozen option knobs.  This test
walks the knobs -- index with / without separators x row-index width x reads staged through LDS or not x look-ahead rows x
pair-shared gathers x PMLs out through the LDS ring x {PML vector, vector + fused bins, bins only} and, for batches of long
reads, the segment plan's K1 / K3 launches --, compares every launch's results with the oracle's (PML vectors, bins,
fast-forward / scan counters), collects the kernel names the library reports having launched (movi_launch_log) and requires
that set to EQUAL the set of pml_kernel_flatp symbols in the shipped code object: an instantiation nothing can reach, or one a
caller can reach that was never compared with the oracle, fails here.
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, classify_py
from movi_amd.engine import masks_of_pml
from test_gpu_parity import mutated_reads, pack

pytestmark = pytest.mark.gpu


def built_walk_kernels():
    """Demangled names of the pml_kernel_flatp kernels in the library's gfx950 code object."""
    import movi_amd
    data = open(movi_amd.lib_path(), "rb").read()
    names = set()
    pos = 0
    tmp = "/tmp/movi_cov_co_%d.o" % os.getpid()
    while True:
        i = data.find(b"__CLANG_OFFLOAD_BUNDLE__", pos)
        if i < 0:
            break
        n = struct.unpack_from("<Q", data, i + 24)[0]
        p = i + 32
        for _ in range(n):
            off, size, ts = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + ts].decode()
            p += ts
            if "gfx950" in triple and size:
                open(tmp, "wb").write(data[i + off:i + off + size])
                syms = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-sW", tmp], capture_output=True, check=True).stdout.decode()
                mangled = [ln.split()[-1] for ln in syms.splitlines() if " FUNC " in ln and "pml_kernel_flatp" in ln]
                dem = subprocess.run(["c++filt"], input="\n".join(mangled).encode(), capture_output=True, check=True).stdout.decode()
                for ln in dem.splitlines():
                    k = ln.strip()
                    if k.startswith("void movi::"):
                        k = k[len("void movi::"):]
                    names.add(k.split(">(")[0] + ">")
        pos = i + 24
    if os.path.exists(tmp):
        os.remove(tmp)
    return names


def take_log():
    from movi_amd._lib import lib
    need = C.c_size_t(0)
    lib().movi_launch_log(None, 0, C.byref(need))               # (switches the log on; clears it)
    return need.value


def read_log():
    from movi_amd._lib import lib
    buf = C.create_string_buffer(1 << 16)
    lib().movi_launch_log(buf, len(buf), None)
    return set(x for x in buf.value.decode().split("\n") if x)


def test_every_built_walk_kernel_is_reachable_and_equals_the_oracle(built_lib, golden_image):
    import torch
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    built = built_walk_kernels()
    assert len(built) == 218, len(built)                        # DESIGN.md section 3 states the count
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(31337)
    short = mutated_reads(rng, ref, 1900, 1, 420) + [b"", b"A", b"N" * 17, ref[:16], ref[100:117], b"ACGT" * 40] + \
        mutated_reads(rng, ref, 90, 500, 1500)                  # mixed lengths, some rolling through the staged stretch
    long_reads = mutated_reads(rng, ref, 40, 1500, 4000)        # segment plan with "seg_len" 256
    dev = torch.device("cuda", 0)
    seen = set()
    take_log()
    for sep in (0, 1):
        img = golden_image(6) if not sep else B.build_index_from_seqs([ref], 6, separators=True)
        gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
        sb, so = pack(short)
        lb, lo = pack(long_reads)
        sexp, sff, ssc = cpu.pml_batch(sb, so, threads=8)
        mexp, mvalid = masks_of_pml(sexp, so)
        lexp, lff, lsc = cpu.pml_batch(lb, lo, threads=8)
        BW, THR = 40, 4
        bins_exp = [classify_py(sexp[int(so[i]):int(so[i + 1])], THR, BW) if len(r) else None for i, r in enumerate(short)]
        n = len(short)
        d_bases = torch.from_numpy(sb.copy()).to(dev)
        d_offs = torch.from_numpy(so.view(np.int64).copy()).to(dev)
        gpu.set_option("classify_fused", 1)
        gpu.set_option("pml_via_mask", 0)                       # (vector launches by the walk's own output paths; the mask launches are asked for by name)
        gpu.set_option("host_masks", 0)
        def check_launches(tag, with_masks):
            """PML vector; reset masks; vector + fused bins, bins only; the segment plan's K1 / K3 -- each against the oracle."""
            nonlocal seen
            # CLS 0: the PML vector
            out, st = gpu.query_pml_packed(sb, so)
            assert (out == sexp).all() and (st.fast_forwards, st.scans, st.errors) == (sff, ssc, 0), tag
            # RING 2: reset masks out (the staged walk writes them itself; ring = 1 changes nothing for it)
            if with_masks:
                words, mst = gpu.query_pml_mask_packed(sb, so)
                assert (words[mvalid] == mexp[mvalid]).all() and (mst.fast_forwards, mst.scans, mst.errors) == (sff, ssc, 0), tag
            # CLS 1 / 2: vector + fused bins, bins only
            for with_vector in (True, False):
                d_out = torch.zeros(max(sb.size, 1), dtype=torch.int16, device=dev)
                d_a = torch.full((n,), -1, dtype=torch.int32, device=dev)
                d_b = torch.full((n,), -1, dtype=torch.int32, device=dev)
                d_s = torch.full((n,), -1, dtype=torch.int64, device=dev)
                gpu.pml_classify_device(d_bases.data_ptr(), d_offs.data_ptr(), n, sb.size, BW, THR,
                                        d_out.data_ptr() if with_vector else 0, d_a.data_ptr(), d_b.data_ptr(), d_s.data_ptr())
                torch.cuda.synchronize()
                a, b, sm = d_a.cpu().numpy(), d_b.cpu().numpy(), d_s.cpu().numpy()
                for i, e in enumerate(bins_exp):
                    if e is None:
                        assert (a[i], b[i], sm[i]) == (0, 0, 0), (tag, i)
                    else:
                        assert (a[i], b[i]) == (e[2], e[3]) and sm[i] == round(e[1] * (e[2] + e[3])), (tag, i)
                if with_vector:
                    assert (d_out.cpu().numpy().view(np.uint16)[:sb.size] == sexp).all(), tag
            # SEG 1 / 2: the segment plan's K1 and K3 launches
            gpu.set_option("seg_len", 256)
            gpu.set_option("seg_probe", 0)
            lout, lst = gpu.query_pml_packed(lb, lo)
            gpu.set_option("seg_len", 2048)
            gpu.set_option("seg_probe", 1)
            assert lst.segments > len(long_reads), tag
            assert (lout == lexp).all() and (lst.fast_forwards, lst.scans, lst.errors) == (lff, lsc, 0), tag
            seen |= read_log()

        for idx64 in (0, 1):
            gpu.set_option("idx64", idx64)
            for stage in (1, 0):
                gpu.set_option("stage_reads", stage)
                for ahead, pair, ring in ([(a, p, r) for a in (0, 1) for p in (0, 1) for r in (0, 1)] if stage else [(0, 0, 0)]):
                    gpu.set_option("ahead_rows", ahead)            # (frees the deep rows the first PML query may have built)
                    gpu.set_option("pair_loads", pair)
                    gpu.set_option("out_ring", ring)
                    check_launches((sep, idx64, stage, ahead, pair, ring), with_masks=bool(stage and not ring))
        # AHD 2: the deep rows (32-bit row indexes, staged reads, no pair-shared gathers)
        gpu.set_option("idx64", 0)
        gpu.set_option("stage_reads", 1)
        gpu.set_option("pair_loads", -1)
        gpu.set_option("deep_rows", 1)
        gpu.set_option("deep", 1)                                  # (whatever the read length: the segment plan's K1 / K3 too)
        for ring in (0, 1):
            gpu.set_option("out_ring", ring)
            check_launches((sep, "deep", ring), with_masks=not ring)
        gpu.close()
        cpu.close()
    seen = set(k for k in seen if k.startswith("pml_kernel_flatp<"))
    unreachable = sorted(built - seen)
    unbuilt = sorted(seen - built)
    assert not unbuilt, unbuilt
    assert not unreachable, ("instantiations no option reaches (prune them, or extend this test): ", unreachable)
