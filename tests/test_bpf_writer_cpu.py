"""CPU suite: the BPF writer of `movi query` (movi_amd/host/output.cpp) -- BpfWriter::append(records), the one-thread loop, and
BpfWriter::append(Chunk, pool), slabs gathered from a chunk's arrays by a pool's threads and written behind it by another thread --
against an independent serialisation of the record format (u16 id_len | id | u64 n | n x u16, src/utils.cpp:202-246): chunks of one
slab, of many slabs, records of every size up to payloads of 1 MiB and more, records larger than a slab, empty records, empty chunks,
shuffled record orders, the two paths alternating on one file."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("bpfw") / "bpf_writer_driver")
    host = os.path.join(ROOT, "movi_amd", "host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "host", "bpf_writer_driver.cpp"),
                           os.path.join(host, "output.cpp"), os.path.join(host, "options.cpp"), os.path.join(host, "reads.cpp"), "-lpthread"])
    return exe


@pytest.mark.parametrize("n,max_len,chunks,big_every", [(0, 10, 1, 0), (1, 0, 1, 0), (50, 300, 1, 0), (200_000, 300, 4, 0), (300_000, 150, 1, 0),
                                                         (3000, 40_000, 3, 0), (400, 20_000, 5, 7), (5, 10, 9, 0)])
@pytest.mark.parametrize("threads", [0, 1, 5])
def test_bpf_writer_bytes(driver, tmp_path, n, max_len, chunks, big_every, threads):
    out = str(tmp_path / "x.bpf")
    subprocess.check_call([driver, out, str(1000 + n), str(n), str(max_len), str(chunks), str(big_every), str(threads)])
    a, b = open(out, "rb").read(), open(out + ".expect", "rb").read()
    assert len(a) == len(b) and a == b


@pytest.mark.parametrize("slab", [64, 4096, 1 << 20])
def test_bpf_writer_small_slabs(driver, tmp_path, slab):
    """Slabs smaller than most records (a slab holds at least one record), and slabs of a few records."""
    out = str(tmp_path / "y.bpf")
    subprocess.check_call([driver, out, "77", "3000", "2500", "4", "0", "3"], env=dict(os.environ, MOVI_BPF_SLAB_BYTES=str(slab)))
    a, b = open(out, "rb").read(), open(out + ".expect", "rb").read()
    assert len(a) == len(b) and a == b
