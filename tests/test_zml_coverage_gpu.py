"""GPU suite (-m gpu): every instantiation of the ZML / count kernels the library was BUILT with -- zml_kernel_flat<MODE, IdxT, SEG, AH,
PSH, CNT> (the lane state machine: ZML parses, the count query, the segment plan's K1) and zml_kernel<MODE, SEG> (base-synchronous: tiny
inputs, A/B, K1 on tables beyond 3 GB, K3) -- is reachable through the options of the C-ABI, and each one is held to the oracle once
(query_zml: /root/reference/src/move_structure_query.cpp:690-785; query_backward_search: src/move_structure_search.cpp:340-352).
The library reports every such launch by name (movi_launch_log); the set seen must EQUAL the set of symbols in the shipped code object
(tests/test_kernel_coverage_gpu.py does the same for the PML walk's 218).  30 kernels: 24 zml_kernel_flat + 6 zml_kernel."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN
from test_gpu_parity import mutated_reads, pack
from test_kernel_coverage_gpu import read_log, take_log

pytestmark = pytest.mark.gpu


def built_zml_kernels():
    import movi_amd
    data = open(movi_amd.lib_path(), "rb").read()
    names, pos = set(), 0
    tmp = "/tmp/movi_zcov_co_%d.o" % os.getpid()
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
                mangled = [ln.split()[-1] for ln in syms.splitlines() if " FUNC " in ln and "zml_kernel" in ln]
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


def test_every_built_zml_and_count_kernel_is_reachable_and_equals_the_oracle(built_lib, golden_image):
    import movi_amd
    from oracle import build_index as B
    from oracle.oracle import Oracle
    built = built_zml_kernels()
    assert len(built) == 30, sorted(built)                      # 24 zml_kernel_flat + 6 zml_kernel
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(4242)
    short = mutated_reads(rng, ref, 900, 1, 400) + [b"", b"A", b"N" * 17, ref[:16], ref[100:117], b"ACGT" * 40, ref[2000:2300]]
    long_reads = mutated_reads(rng, ref, 30, 1500, 4000)
    sb, so = pack(short)
    lb, lo = pack(long_reads)
    seen = set()
    take_log()
    for kmode, img in ((6, golden_image(6)), (3, B.build_index_from_seqs([ref], 3))):
        gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
        zexp = cpu.zml_batch(sb, so, threads=8)
        lzexp = cpu.zml_batch(lb, lo, threads=8)
        em, ec = cpu.count_batch(sb, so, threads=8)
        for idx64 in (0, 1):
            gpu.set_option("idx64", idx64)
            T = "unsigned long" if idx64 else "unsigned int"
            # ZML: the state machine (own loads / pair-shared gathers / on the look-ahead rows), the base-synchronous kernel
            for variant, pair, ahead in ((1, 0, 0), (1, 1, 0), (1, 0, 1), (0, 0, 0)):
                if ahead and kmode != 6:
                    continue
                gpu.set_option("zml_variant", variant)
                gpu.set_option("pair_loads", pair)
                if kmode == 6:
                    gpu.set_option("ahead_rows", ahead)
                gpu.set_option("zml_ahead", ahead)
                z, st = gpu.query_zml_packed(sb, so)
                want = "zml_kernel_flat<%d, %s, 0, %d, %d, 0>" % (kmode, T, ahead, pair) if variant else "zml_kernel<%d, 0>" % kmode
                assert gpu.last_launch()["kernel"] == want, (gpu.last_launch(), want)
                assert (z == zexp).all() and st.errors == 0, (kmode, idx64, variant, pair, ahead)
            gpu.set_option("zml_variant", -1)
            gpu.set_option("zml_ahead", 0)
            gpu.set_option("pair_loads", -1)
            # the segment plan: K1 as the state machine (K3: zml_kernel<M, 2>)
            gpu.set_option("seg_len", 256)
            gpu.set_option("seg_probe", 0)
            z, st = gpu.query_zml_packed(lb, lo)
            gpu.set_option("seg_len", 2048)
            gpu.set_option("seg_probe", 1)
            assert st.segments > len(long_reads) and (z == lzexp).all(), (kmode, idx64)
            assert gpu.last_launch()["kernel"] == "zml_kernel_flat<%d, %s, 1, 0, 0, 0>" % (kmode, T)
            # the count query on the state machine
            gpu.set_option("count_variant", 1)
            for pair in (0, 1):
                gpu.set_option("pair_loads", pair)
                m, c, st = gpu.query_count_packed(sb, so)
                assert gpu.last_launch()["kernel"] == "zml_kernel_flat<%d, %s, 0, 0, %d, 1>" % (kmode, T, pair)
                assert (m == em).all() and (c == ec).all() and st.errors == 0, (kmode, idx64, pair)
            if kmode == 6:                                      # ... and on the look-ahead rows ("zml_ahead" 1; round 6)
                gpu.set_option("pair_loads", 0)
                gpu.set_option("ahead_rows", 1)
                gpu.set_option("zml_ahead", 1)
                m, c, st = gpu.query_count_packed(sb, so)
                assert gpu.last_launch()["kernel"] == "zml_kernel_flat<6, %s, 0, 1, 0, 1>" % T and gpu.last_launch()["ahead"] == 1
                assert (m == em).all() and (c == ec).all() and st.errors == 0, (kmode, idx64, "ahead")
                gpu.set_option("zml_ahead", 0)
            gpu.set_option("count_variant", -1)
            gpu.set_option("pair_loads", -1)
            seen |= read_log()
        gpu.set_option("idx64", 0)
        gpu.close()
        cpu.close()
    # zml_kernel<M, 1>: K1 of the segment plan where the table is beyond the state machine's 3 GB (and its pairs are off): reached here
    # through the same launcher on a table of fewer than 8 rows (the state machine needs two windows of rows)
    for kmode in (6, 3):
        img = B.build_index_from_seqs([b"ACG"], kmode)
        gpu, cpu = movi_amd.MoveIndex.from_image(img), Oracle(img)
        if gpu.desc.r < 8:
            reads = [b"ACGT" * 200, b"CGTA" * 150, b"A" * 700] * 4
            b2, o2 = pack(reads)
            gpu.set_option("seg_len", 64)
            gpu.set_option("seg_probe", 0)
            z, st = gpu.query_zml_packed(b2, o2)
            assert (z == cpu.zml_batch(b2, o2, threads=2)).all(), kmode
            seen |= read_log()
        gpu.close()
        cpu.close()
    seen = set(k for k in seen if k.startswith("zml_kernel"))
    unreachable = sorted(built - seen)
    unbuilt = sorted(seen - built)
    assert not unbuilt, unbuilt
    assert not unreachable, ("instantiations no option reaches (prune them, or extend this test): ", unreachable)
