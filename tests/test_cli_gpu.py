"""GPU suite for the `movi` host binary: the reference's own CLI-level golden test
(tests/test_pml.cpp:6-55: query --stdout | LC_ALL=C sort | diff golden), BPF bytes,
`view`, count lines, classification report / filter, --reverse, --ignore-illegal-chars,
stdin input -- all against the oracle."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, classify_py, read_fastx, stdout_line

pytestmark = pytest.mark.gpu
MOVI = os.path.join(ROOT, "movi_amd", "bin", "movi")
IDX = {6: os.path.join(GOLDEN, "index_regular-thresholds"), 8: os.path.join(GOLDEN, "index_blocked-thresholds")}
TYPE = {6: "regular-thresholds", 8: "blocked-thresholds"}


@pytest.fixture(scope="module")
def movi_bin(built_lib):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "movi_amd", "csrc")], stdout=subprocess.DEVNULL)
    return MOVI


@pytest.fixture(scope="module")
def oracles(golden_image):
    from oracle.oracle import Oracle
    return {m: Oracle(golden_image(m)) for m in (6, 8)}


def run(args, **kw):
    return subprocess.run([MOVI] + args, capture_output=True, **kw)


def plan_order(reads_path, flags):
    r = run(["plan", "-r", reads_path] + flags)
    assert r.returncode == 0
    out = []
    for l in r.stdout.split(b"\n"):
        if l:
            rest = l.split(b"\t", 1)[1]
            out.append(rest.rsplit(b"\t", 1)[0])
    return out


# the reference's CLI golden test matrix restricted to the modes in scope (tests/test_pml.cpp:57-105)
@pytest.mark.parametrize("mode", [6, 8])
@pytest.mark.parametrize("flags", [["--no-prefetch", "-t1"], ["-s16", "-t1"], ["-s4", "-t4"], ["--no-prefetch", "-t4"]])
@pytest.mark.parametrize("reads", ["sample.fastq", "sample.fasta"])
def test_reference_cli_golden(movi_bin, mode, flags, reads):
    r = run(["query", "--index", IDX[mode], "--read", os.path.join(GOLDEN, reads), "--pml"] + flags + ["--stdout"])
    assert r.returncode == 0, r.stderr
    got = b"".join(sorted(r.stdout.splitlines(keepends=True)))          # LC_ALL=C sort
    assert got == open(os.path.join(GOLDEN, "sample.fastq.pmls.sorted"), "rb").read()


def pack(reads):
    offs = np.zeros(len(reads) + 1, np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    return np.frombuffer(b"".join(reads), np.uint8).copy(), offs


def write_mixed_reads(path, rng, ref, n=300):
    recs = []
    with open(path, "wb") as f:
        for i in range(n):
            L = int(rng.integers(1, 900))
            s = int(rng.integers(0, len(ref) - L))
            seq = bytearray(ref[s:s + L])
            for k in range(L):
                if rng.random() < 0.02:
                    seq[k] = b"ACGTNacgt"[rng.integers(0, 9)]
            hdr = b">m%d" % i + [b"", b" desc", b"\tz"][i % 3]
            f.write(hdr + b"\n")
            for j in range(0, L, 70):
                f.write(bytes(seq[j:j + 70]) + b"\n")
            rid = hdr[1:] if i % 3 == 0 else hdr[1:hdr.index(b" " if i % 3 == 1 else b"\t") + 1]
            recs.append((rid, bytes(seq)))
    return recs


@pytest.mark.parametrize("mode", [6, 8])
def test_bpf_bytes_view_and_order(movi_bin, oracles, tmp_path, mode):
    from oracle import build_index as B
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    reads_path = str(tmp_path / "mixed.fa")
    recs = write_mixed_reads(reads_path, np.random.default_rng(5 + mode), ref)
    by_id = dict(recs)
    prefix = str(tmp_path / "out")
    r = run(["query", "-i", IDX[mode], "-r", reads_path, "-o", prefix, "-s8", "-t1"])
    assert r.returncode == 0, r.stderr
    blob = open(prefix + ".pml.bpf", "rb").read()
    exp = struct.pack("<IBBBBHxx", 0x42504600, 1, 0, 0, 16, 0)
    for rid in plan_order(reads_path, ["-s8"]):                         # record order = strand scheduler order
        p = oracles[mode].pml(by_id[rid])
        exp += struct.pack("<H", len(rid)) + rid + struct.pack("<Q", len(p)) + p.astype("<u2").tobytes()
    assert blob == exp
    # default file name: <read_file>.<type>.pml.bpf (src/utils.cpp:346-356)
    r = run(["query", "-i", IDX[mode], "-r", reads_path, "-n"])
    assert r.returncode == 0
    default_name = reads_path + "." + TYPE[mode] + ".pml.bpf"
    assert os.path.exists(default_name)
    # view == --stdout (file order with --no-prefetch)
    v = run(["view", "--bpf", default_name])
    s = run(["query", "-i", IDX[mode], "-r", reads_path, "-n", "--stdout"])
    assert v.returncode == 0 and s.returncode == 0 and v.stdout == s.stdout
    assert s.stdout == b"".join(b">" + rid + b"\n" + stdout_line(oracles[mode].pml(seq)).encode() + b"\n" for rid, seq in recs)


def test_bpf_bytes_with_pooled_order_and_many_slabs(movi_bin, oracles, tmp_path):
    """Round 5: 9 000 ragged reads -- enough for the writer stage's pool to compute the strand-scheduler order by ranges of batches --
    and slabs of 64 KiB, so that the BPF file leaves through hundreds of ring slots: the bytes are the oracle's PMLs in `movi plan`'s
    order, and the one-thread loop (MOVI_BPF_SERIAL=1) writes the same file."""
    from oracle import build_index as B
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    reads_path = str(tmp_path / "many.fa")
    recs = write_mixed_reads(reads_path, np.random.default_rng(77), ref, n=9000)
    by_id = dict(recs)
    assert len(by_id) == 9000
    prefix = str(tmp_path / "out")
    r = run(["query", "-i", IDX[6], "-r", reads_path, "-o", prefix, "-s16", "-t1"], env=dict(os.environ, MOVI_BPF_SLAB_BYTES="65536"))
    assert r.returncode == 0, r.stderr
    blob = open(prefix + ".pml.bpf", "rb").read()
    order = plan_order(reads_path, ["-s16"])
    assert order == [x for x in plan_order(reads_path, ["-s16"])] and sorted(order) == sorted(by_id)
    bases, offs = pack([by_id[rid] for rid in order])
    pml, _, _ = oracles[6].pml_batch(bases, offs, threads=4)
    exp = [struct.pack("<IBBBBHxx", 0x42504600, 1, 0, 0, 16, 0)]
    for k, rid in enumerate(order):
        p = pml[int(offs[k]):int(offs[k + 1])]
        exp.append(struct.pack("<H", len(rid)) + rid + struct.pack("<Q", len(p)) + p.astype("<u2").tobytes())
    assert blob == b"".join(exp)
    r2 = run(["query", "-i", IDX[6], "-r", reads_path, "-o", prefix + "2", "-s16", "-t1"], env=dict(os.environ, MOVI_BPF_SERIAL="1"))
    assert r2.returncode == 0 and open(prefix + "2.pml.bpf", "rb").read() == blob


@pytest.mark.parametrize("mode", [6, 8])
def test_count_lines(movi_bin, oracles, tmp_path, mode):
    from oracle import build_index as B
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    reads_path = str(tmp_path / "c.fa")
    recs = write_mixed_reads(reads_path, np.random.default_rng(50 + mode), ref, n=200)
    r = run(["query", "-i", IDX[mode], "-r", reads_path, "--count", "--no-prefetch", "--stdout"])
    assert r.returncode == 0, r.stderr
    exp = b""
    for rid, seq in recs:
        m, c = oracles[mode].count(seq)
        exp += rid + b"\t%d/%d\t%d\n" % (m, len(seq), c)
    assert r.stdout == exp
    # file output + prefetch-mode order: same multiset of lines
    r = run(["query", "-i", IDX[mode], "-r", reads_path, "--count", "-o", str(tmp_path / "cc")])
    assert r.returncode == 0
    got = open(str(tmp_path / "cc") + ".count.matches", "rb").read()
    assert sorted(got.splitlines()) == sorted(exp.splitlines())


@pytest.mark.parametrize("mode", [6, 8])
def test_zml_cli(movi_bin, oracles, tmp_path, mode):
    """`movi query --zml`: <prefix>.zml.bpf (src/utils.cpp:346-356, query_type "zml"), --stdout, view."""
    from oracle import build_index as B
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    reads_path = str(tmp_path / "z.fa")
    recs = write_mixed_reads(reads_path, np.random.default_rng(70 + mode), ref, n=200)
    s = run(["query", "-i", IDX[mode], "-r", reads_path, "--zml", "-n", "--stdout"])
    assert s.returncode == 0, s.stderr
    assert s.stdout == b"".join(b">" + rid + b"\n" + stdout_line(oracles[mode].zml(seq)).encode() + b"\n" for rid, seq in recs)
    r = run(["query", "-i", IDX[mode], "-r", reads_path, "--zml", "-n"])
    assert r.returncode == 0, r.stderr
    name = reads_path + "." + TYPE[mode] + ".zml.bpf"
    blob = open(name, "rb").read()
    exp = struct.pack("<IBBBBHxx", 0x42504600, 1, 0, 0, 16, 0)
    for rid, seq in recs:
        z = oracles[mode].zml(seq)
        exp += struct.pack("<H", len(rid)) + rid + struct.pack("<Q", len(z)) + z.astype("<u2").tobytes()
    assert blob == exp
    v = run(["view", "--bpf", name])
    assert v.returncode == 0 and v.stdout == s.stdout
    # prefetch mode: same records, scheduler order (a permutation); --pml given after --zml wins
    r = run(["query", "-i", IDX[mode], "-r", reads_path, "--zml", "-s8", "-o", str(tmp_path / "zz"), "--gpus", "1"])
    assert r.returncode == 0, r.stderr
    v2 = run(["view", "--bpf", str(tmp_path / "zz") + ".zml.bpf"])
    assert sorted(v2.stdout.split(b">")) == sorted(s.stdout.split(b">"))
    r = run(["query", "-i", IDX[mode], "-r", reads_path, "--zml", "--pml", "-n", "--stdout"])
    assert r.stdout == b"".join(b">" + rid + b"\n" + stdout_line(oracles[mode].pml(seq)).encode() + b"\n" for rid, seq in recs)


def test_classify_report_and_filter(movi_bin, oracles, tmp_path):
    from oracle import build_index as B
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    idx = str(tmp_path / "idx")
    shutil.copytree(IDX[6], idx)
    # movi.pml.nulldb: u64 num | f64 mean | u64 percentile | u64 stats[num]  (emperical_null_database.cpp:95-126)
    with open(os.path.join(idx, "movi.pml.nulldb"), "wb") as f:
        f.write(struct.pack("<QdQ", 3, 4.5, 6) + struct.pack("<3Q", 4, 5, 6))
    thr = max(6, 3) + 1
    rng = np.random.default_rng(77)
    reads_path = str(tmp_path / "cl.fa")
    recs = []
    with open(reads_path, "wb") as f:
        for i in range(60):
            L = int(rng.integers(100, 1500))
            if i % 2:
                s = int(rng.integers(0, len(ref) - L)); seq = ref[s:s + L]
            else:
                seq = bytes(rng.choice(list(b"ACGT"), size=L).astype(np.uint8))
            f.write(b">c%d\n" % i + seq + b"\n")
            recs.append((b"c%d" % i, seq))
    r = run(["query", "-i", idx, "-r", reads_path, "--classify", "--stdout", "-n"])
    assert r.returncode == 0, r.stderr
    lines = r.stdout.decode().split("\n")
    assert lines[0] == "%-30s%-15s%-19s%-2d%-5s%-12s%-12s" % ("read id:", "status:", "avg max-value (thr=", thr, "):",
                                                              "above thr:", "below thr:")
    found_ids = []
    for (rid, seq), line in zip(recs, lines[1:]):
        found, avg, above, below = classify_py(oracles[6].pml(seq), thr)
        exp = "%-30s%-15s%-26s%-12d%-12d" % (rid.decode(), "FOUND" if found else "NOT_PRESENT", "%.3g" % avg, above, below)
        assert line == exp
        if found:
            found_ids.append(rid)
    assert 0 < len(found_ids) < len(recs)
    # report file + bpf when not --stdout
    r = run(["query", "-i", idx, "-r", reads_path, "--classify", "-n"])
    assert r.returncode == 0
    rep = open(reads_path + ".regular-thresholds.pml.report").read()
    assert rep == r"".join(l + "\n" for l in lines[:-1])
    assert os.path.exists(reads_path + ".regular-thresholds.pml.bpf")
    # --filter echoes FOUND reads, -v the others (src/read_processor.cpp:569-576, utils.cpp:291-294)
    r = run(["query", "-i", idx, "-r", reads_path, "--classify", "--filter", "-n"])
    assert r.stdout == b"".join(b">" + rid + b"\n" + seq + b"\n" for rid, seq in recs if rid in found_ids)
    r = run(["query", "-i", idx, "-r", reads_path, "--classify", "--filter", "-v", "-n"])
    assert r.stdout == b"".join(b">" + rid + b"\n" + seq + b"\n" for rid, seq in recs if rid not in found_ids)


def null_stats_py(values):
    """EmpNullDatabase::compute_stats, src/emperical_null_database.cpp:47-92."""
    v = np.asarray(values, np.uint64)
    uniq, cnt = np.unique(v, return_counts=True)
    common = uniq[cnt >= 5]
    perc = int(common.max()) if common.size else 0
    return struct.pack("<QdQ", v.size, float(v.astype(np.float64).sum() / v.size), perc) + v.astype("<u8").tobytes(), perc


@pytest.mark.parametrize("mode", [6, 8])
def test_null_database_and_reference_filter_golden(movi_bin, oracles, tmp_path, mode):
    """`movi null --gen-reads` (parse_null_reads + generate_null_statistics) and the reference's own
    classification golden: tests/test_classification.cpp:54-100 runs `query --pml --filter --invert --stdout`
    on sample.fasta and diffs the sorted output against sample.fasta.pmls.filtered_notfound.sorted."""
    from oracle import build_index as B
    idx = str(tmp_path / "idx")
    shutil.copytree(IDX[mode], idx)
    ref_path = os.path.join(GOLDEN, "ref.fasta")
    env = dict(os.environ, MOVI_NULL_SEED="7")
    r = run(["null", "-i", idx, "--gen-reads", "-f", ref_path], env=env)
    assert r.returncode == 0, r.stderr
    recs = B.read_fasta(ref_path)
    nulls = read_fastx(os.path.join(idx, "null_reads.fasta"))
    assert len(nulls) == 100 * len(recs) and all(len(s) == 150 for _, s in nulls)
    assert [i for i, _ in nulls] == [b"read_%d" % k for k in range(len(nulls))]
    raw = open(ref_path, "rb").read().split(b"\n")
    whole = b"".join(l for l in raw if not l.startswith(b">"))
    assert all(s[::-1] in whole for _, s in nulls)                  # reversed chunks of the reference
    vals = np.concatenate([oracles[mode].pml(s) for _, s in nulls])
    blob, perc = null_stats_py(vals)
    assert open(os.path.join(idx, "movi.pml.nulldb"), "rb").read() == blob
    assert 4 <= perc <= 40
    # same seed -> same reads; ZML database from the existing reads (no --gen-reads)
    r = run(["null", "-i", idx, "--zml"], env=env)
    assert r.returncode == 0, r.stderr
    zvals = np.concatenate([oracles[mode].zml(s) for _, s in nulls])
    assert open(os.path.join(idx, "movi.zml.nulldb"), "rb").read() == null_stats_py(zvals)[0]
    # the reference's golden (all 25 simulated reads are foreign to ref.fasta)
    gold = open(os.path.join(GOLDEN, "sample.fasta.pmls.filtered_notfound.sorted"), "rb").read()
    for flags in (["-s16", "-t1"], ["--no-prefetch", "-t1"]):
        q = run(["query", "--index", idx, "--read", os.path.join(GOLDEN, "sample.fasta"), "--pml", "--filter", "--invert"]
                + flags + ["--stdout"])
        assert q.returncode == 0, q.stderr
        assert b"".join(sorted(q.stdout.splitlines(keepends=True))) == gold
    q = run(["query", "--index", idx, "--read", os.path.join(GOLDEN, "sample.fasta"), "--pml", "--filter", "--stdout"])
    assert q.returncode == 0 and q.stdout == b""
    # reads drawn from the reference itself are FOUND; ZML classification uses movi.zml.nulldb
    own = tmp_path / "own.fa"
    seq = recs[0][1]
    own.write_bytes(b"".join(b">o%d\n%s\n" % (k, seq[k * 700:k * 700 + 450]) for k in range(8)))
    for qt in ("--pml", "--zml"):
        q = run(["query", "-i", idx, "-r", str(own), qt, "--filter", "--stdout", "-n"])
        assert q.returncode == 0 and q.stdout == own.read_bytes()


def test_reverse_illegal_chars_stdin_and_errors(movi_bin, oracles, tmp_path):
    reads = read_fastx(os.path.join(GOLDEN, "sample.fastq"))
    r = run(["query", "-i", IDX[6], "-r", os.path.join(GOLDEN, "sample.fastq"), "--reverse", "-n", "--stdout"])
    assert r.stdout == b"".join(b">" + i + b"\n" + stdout_line(oracles[6].pml(s[::-1])).encode() + b"\n" for i, s in reads)
    p = tmp_path / "n.fa"
    p.write_bytes(b">xy\nACGTNNACGTacgtACGT\n")
    r = run(["query", "-i", IDX[6], "-r", str(p), "--ignore-illegal-chars", "1", "-n", "--stdout"])
    assert r.stdout == b">xy\n" + stdout_line(oracles[6].pml(b"ACGTAAACGTAAAAACGT")).encode() + b"\n"
    data = open(os.path.join(GOLDEN, "sample.fasta"), "rb").read()
    r = run(["query", "-i", IDX[8], "-r", "-", "-n", "--stdout"], input=data)
    assert r.returncode == 0
    assert r.stdout == b"".join(b">" + i + b"\n" + stdout_line(oracles[8].pml(s)).encode() + b"\n" for i, s in reads)
    r = run(["query", "-i", str(tmp_path / "nope"), "-r", str(p)])
    assert r.returncode == 1 and b"Failed to open the index file" in r.stderr
    r = run(["query", "-i", IDX[6], "-r", str(tmp_path / "nope.fa")])
    assert r.returncode == 1 and b"does not exist" in r.stderr
    # --no-output: nothing is written, exit code 0
    q = tmp_path / "q.fa"
    q.write_bytes(b">xy\nACGT\n")
    r = run(["query", "-i", IDX[6], "-r", str(q), "--no-output"])
    assert r.returncode == 0 and r.stdout == b"" and not os.path.exists(str(q) + ".regular-thresholds.pml.bpf")


def test_multi_gpu_sharding_path(movi_bin, tmp_path):
    """`--gpus N`: the index is read once and uploaded per GPU, each chunk of reads is sharded by bases over
    N host threads / handles.  On a 1-GPU box MOVI_SHARE_GPU=1 maps every logical GPU to device 0; the
    output must be byte-identical to the single-GPU run (PML file order and count)."""
    from oracle import build_index as B
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    reads_path = str(tmp_path / "mg.fa")
    write_mixed_reads(reads_path, np.random.default_rng(9), ref, n=500)
    env = dict(os.environ, MOVI_SHARE_GPU="1")
    for extra in (["--stdout"], ["--count", "--stdout", "-n"]):
        one = run(["query", "-i", IDX[8], "-r", reads_path] + extra)
        many = run(["query", "-i", IDX[8], "-r", reads_path, "--gpus", "3"] + extra, env=env)
        assert one.returncode == 0 and many.returncode == 0, many.stderr
        assert one.stdout == many.stdout and len(one.stdout) > 1000


@pytest.mark.parametrize("mode", [6, 8])
def test_explicit_gpus_flag_replicates_through_rccl(movi_bin, tmp_path, mode):
    """`--gpus N` given explicitly (N = 1 on this box): the index goes through movi_index_load_replicated -- one upload,
    one RCCL broadcast (a communicator of one rank here), per-GPU expansion -- instead of movi_index_load.  Same bytes out;
    asking for more GPUs than there are is an error, not a silent fallback; --no-output walks without fetching anything."""
    from oracle import build_index as B
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    reads_path = str(tmp_path / "rg.fa")
    write_mixed_reads(reads_path, np.random.default_rng(19), ref, n=400)
    one = run(["query", "-i", IDX[mode], "-r", reads_path, "-o", str(tmp_path / "a")])
    rep = run(["query", "-i", IDX[mode], "-r", reads_path, "-o", str(tmp_path / "b"), "--gpus", "1"])
    assert one.returncode == 0 and rep.returncode == 0, rep.stderr
    a = open(str(tmp_path / "a") + ".pml.bpf", "rb").read()
    assert a == open(str(tmp_path / "b") + ".pml.bpf", "rb").read() and len(a) > 100000
    import torch
    too_many = run(["query", "-i", IDX[mode], "-r", reads_path, "--no-output", "--gpus", str(torch.cuda.device_count() + 1)])
    assert too_many.returncode == 1 and b"visible" in too_many.stderr
    quiet = run(["query", "-i", IDX[mode], "-r", reads_path, "--no-output", "--verbose"])
    assert quiet.returncode == 0 and quiet.stdout == b"" and b"400 reads are processed" in quiet.stderr
    assert b"Stage times: parse" in quiet.stderr


def test_logs_files(movi_bin, oracles, tmp_path):
    """`movi query --logs`: <prefix>.scans / .fastforwards hold one `>id` line and one line of space-terminated per-base values
    per read (output_logs, src/utils.cpp:268-289), in the BPF file's record order; .costs (a CPU strand's nanoseconds) is
    written as zeros."""
    from oracle import build_index as B
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    reads_path = str(tmp_path / "lg.fa")
    recs = dict(write_mixed_reads(reads_path, np.random.default_rng(23), ref, n=120))
    prefix = str(tmp_path / "lg")
    r = run(["query", "-i", IDX[6], "-r", reads_path, "-o", prefix, "--logs", "-s8", "-t1"])
    assert r.returncode == 0, r.stderr
    order = plan_order(reads_path, ["-s8", "-t1"])
    files = {k: open(prefix + ".pml." + k, "rb").read().split(b"\n") for k in ("costs", "scans", "fastforwards")}
    for k, lines in files.items():
        assert lines[-1] == b"" and len(lines) == 2 * len(order) + 1, k
        assert [l[1:] for l in lines[0:-1:2]] == order, k
    for j, rid in enumerate(order):
        eo, ef, es = oracles[6].pml_logs(recs[rid])
        assert files["scans"][2 * j + 1] == b"".join(b"%d " % v for v in es)
        assert files["fastforwards"][2 * j + 1] == b"".join(b"%d " % v for v in ef)
        assert files["costs"][2 * j + 1] == b"0 " * len(eo)
    # the PML file of a --logs run is the ordinary one
    plain = run(["query", "-i", IDX[6], "-r", reads_path, "-o", str(tmp_path / "pl"), "-s8", "-t1"])
    assert plain.returncode == 0
    assert open(prefix + ".pml.bpf", "rb").read() == open(str(tmp_path / "pl") + ".pml.bpf", "rb").read()


@pytest.mark.parametrize("mode", [6, 8])
def test_separators_index_through_the_cli(movi_bin, tmp_path, mode):
    """`movi query` on a `movi build --separators` index (reference KAT sizes tests/test_build.cpp:79,95):
    --pml --stdout and --count lines against the oracle; '%' in a read is an illegal character."""
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    img = B.build_index_from_seqs([ref], mode, separators=True)
    d = tmp_path / "sep_index"
    d.mkdir()
    (d / "index.movi").write_bytes(img)
    cpu = Oracle(img)
    rng = np.random.default_rng(55 + mode)
    reads_path = str(tmp_path / "mixed.fa")
    recs = write_mixed_reads(reads_path, rng, ref, n=120)
    with open(reads_path, "ab") as f:
        f.write(b">withsep\nACGTAC%GTACGT\n")
    recs.append((b"withsep", b"ACGTAC%GTACGT"))
    r = run(["query", "--index", str(d), "--read", reads_path, "--pml", "--no-prefetch", "-t1", "--stdout"])
    assert r.returncode == 0, r.stderr
    assert b"regular-thresholds" in r.stderr or b"blocked-thresholds" in r.stderr
    exp = b"".join(b">" + rid + b"\n" + stdout_line(cpu.pml(seq)).encode() + b"\n" for rid, seq in recs)
    assert r.stdout == exp
    r = run(["query", "--index", str(d), "--read", reads_path, "--count", "--no-prefetch", "-t1", "--stdout"])
    assert r.returncode == 0, r.stderr
    exp = b""
    for rid, seq in recs:
        m, c = cpu.count(seq)
        exp += rid + b"\t%d/%d\t%d\n" % (m, len(seq), c)
    assert r.stdout == exp


# tests/test_pml.cpp:66-68, :98-100 of the reference: the sampled-thresholds index against the same golden file
@pytest.mark.parametrize("flags", [["--no-prefetch", "-t1"], ["-s4", "-t1"]])
def test_reference_cli_golden_sampled_thresholds(movi_bin, tmp_path, flags):
    from oracle import build_index as B
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    d = tmp_path / "sampled"
    d.mkdir()
    (d / "index.movi").write_bytes(B.build_index_from_seqs([ref], 7))
    r = run(["query", "--index", str(d), "--read", os.path.join(GOLDEN, "sample.fastq"), "--pml"] + flags + ["--stdout"])
    assert r.returncode == 0, r.stderr
    assert b"sampled-thresholds" in r.stderr
    got = b"".join(sorted(r.stdout.splitlines(keepends=True)))          # LC_ALL=C sort
    assert got == open(os.path.join(GOLDEN, "sample.fastq.pmls.sorted"), "rb").read()
    # file outputs carry the index type in their names (src/utils.cpp:348-362)
    reads = tmp_path / "s.fastq"
    shutil.copy(os.path.join(GOLDEN, "sample.fastq"), reads)
    assert run(["query", "--index", str(d), "--read", str(reads), "--pml"]).returncode == 0
    assert os.path.exists(str(reads) + ".sampled-thresholds.pml.bpf")
    assert run(["query", "--index", str(d), "--read", str(reads), "--count"]).returncode == 0
    assert os.path.exists(str(reads) + ".sampled-thresholds.count.matches")


@pytest.mark.parametrize("mode,name", [(3, "regular"), (2, "blocked")])
def test_cli_regular_and_blocked_indexes(movi_bin, tmp_path, mode, name):
    """`movi query` on the threshold-less index types: --count and --zml served (file names carry the type, src/utils.cpp:20-28),
    --pml refused with the reason (the reference would reposition randomly)."""
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    d = tmp_path / name
    d.mkdir()
    img = B.build_index_from_seqs([ref], mode)
    (d / "index.movi").write_bytes(img)
    reads = tmp_path / "s.fastq"
    shutil.copy(os.path.join(GOLDEN, "sample.fastq"), reads)
    r = run(["query", "--index", str(d), "--read", str(reads), "--count", "--no-prefetch", "--stdout"])
    assert r.returncode == 0, r.stderr
    assert (b"The " + name.encode() + b" index") in r.stderr
    cpu = Oracle(img)
    exp = b""
    for rid, seq in read_fastx(str(reads)):
        m, c = cpu.count(seq)
        exp += rid + b"\t%d/%d\t%d\n" % (m, len(seq), c)
    assert r.stdout == exp
    assert run(["query", "--index", str(d), "--read", str(reads), "--zml"]).returncode == 0
    assert os.path.exists(str(reads) + "." + name + ".zml.bpf")
    r = run(["query", "--index", str(d), "--read", str(reads), "--pml"])
    assert r.returncode == 1 and b"thresholds" in r.stderr


def test_movi_build_then_query_reproduces_golden(movi_bin, tmp_path):
    """End to end inside this CLI: `movi build` from the reference's ref.fasta, then `movi query --pml --stdout` on its
    sample.fastq reproduces the reference's golden PML file (tests/test_pml.cpp:89-105)."""
    d = tmp_path / "built"
    assert run(["build", "-i", str(d), "-f", os.path.join(GOLDEN, "ref.fasta")]).returncode == 0
    r = run(["query", "--index", str(d), "--read", os.path.join(GOLDEN, "sample.fastq"), "--pml", "--no-prefetch", "--stdout"])
    assert r.returncode == 0, r.stderr
    got = b"".join(sorted(r.stdout.splitlines(keepends=True)))          # LC_ALL=C sort
    assert got == open(os.path.join(GOLDEN, "sample.fastq.pmls.sorted"), "rb").read()


def test_stdout_and_view_on_a_multi_chunk_file(movi_bin, tmp_path):
    """40 k x 150 bp reads (6 Mbases: past the 4 Mbase threshold of the threaded `--stdout` formatter) with mixed-length
    stragglers: `query --stdout` (prefetch order and file order), `query` + `view` of the BPF file and the oracle agree."""
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(8)
    lines, reads = [], []
    for i in range(40000):
        L = 150 if i % 50 else int(rng.integers(1, 600))
        st = int(rng.integers(0, len(ref) - L))
        r = bytearray(ref[st:st + L])
        for k in rng.integers(0, L, max(1, L // 70)):
            r[k] = b"ACGTN"[rng.integers(0, 5)]
        reads.append(bytes(r))
        lines += [b">q%d" % i, bytes(r)]
    path = tmp_path / "many.fa"
    path.write_bytes(b"\n".join(lines) + b"\n")
    cpu = Oracle(open(os.path.join(IDX[6], "index.movi"), "rb").read())
    exp = {b"q%d" % i: stdout_line(cpu.pml(r)).encode() for i, r in enumerate(reads)}
    file_order = b"".join(b">q%d\n" % i + exp[b"q%d" % i] + b"\n" for i in range(len(reads)))
    a = run(["query", "-i", IDX[6], "-r", str(path), "-n", "--stdout"])
    assert a.returncode == 0 and a.stdout == file_order
    b = run(["query", "-i", IDX[6], "-r", str(path), "--stdout"])                 # strand-scheduler order: a permutation
    assert b.returncode == 0 and sorted(b.stdout.split(b">")) == sorted(file_order.split(b">"))
    assert run(["query", "-i", IDX[6], "-r", str(path), "-n", "-o", str(tmp_path / "o")]).returncode == 0
    v = run(["view", "--bpf", str(tmp_path / "o.pml.bpf")])
    assert v.returncode == 0 and v.stdout == file_order


def test_cli_accepts_mmap_flag(movi_bin):
    """`movi query --mmap` (the reference's "use memory mapping to read the index"): accepted, same bytes -- the index file
    is always mapped and its rows uploaded from the page cache (movi_index_load)."""
    base = ["query", "--index", IDX[6], "--read", os.path.join(GOLDEN, "sample.fastq"), "--pml", "--no-prefetch", "--stdout"]
    a, b = run(base), run(base + ["--mmap"])
    assert a.returncode == 0 and b.returncode == 0, (a.stderr, b.stderr)
    assert a.stdout == b.stdout and len(a.stdout) > 0


@pytest.mark.parametrize("shape", ["long", "short_with_a_giant"])
def test_bpf_writer_paths_for_long_records(movi_bin, tmp_path, shape):
    """The BPF writer copies records into an 8 MiB buffer and writes payloads of a MiB and more straight from the result
    array: long records only, and a giant one in the middle of a buffer of short ones -- byte for byte against the oracle
    (file order)."""
    from oracle import build_index as B
    from oracle.oracle import Oracle
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    rng = np.random.default_rng(77 + len(shape))
    def piece(L):
        out = bytearray()
        while len(out) < L:
            st = int(rng.integers(0, len(ref) - 1000))
            out += ref[st: st + min(int(rng.integers(500, 50000)), L - len(out))]
        for k in rng.integers(0, L, max(1, L // 200)):
            out[k] = b"ACGTN"[rng.integers(0, 5)]
        return bytes(out)
    if shape == "long":
        lens = [int(x) for x in rng.integers(3000, 30000, 150)] + [700_000, 2100, 5]
    else:
        lens = [int(x) for x in rng.integers(1, 300, 30000)]
        lens[12345] = 600_000                                          # 1.2 MB of PMLs in the middle of a slab
    reads = [piece(L) for L in lens]
    path = tmp_path / "reads.fa"
    path.write_bytes(b"".join(b">r%d some comment\n" % i + r + b"\n" for i, r in enumerate(reads)))
    prefix = str(tmp_path / "o")
    r = run(["query", "-i", IDX[6], "-r", str(path), "-n", "-o", prefix])
    assert r.returncode == 0, r.stderr
    cpu = Oracle(open(os.path.join(IDX[6], "index.movi"), "rb").read())
    bases, offs = np.frombuffer(b"".join(reads), np.uint8), np.zeros(len(reads) + 1, np.uint64)
    np.cumsum([len(x) for x in reads], out=offs[1:])
    pml, _, _ = cpu.pml_batch(bases, offs, threads=8)
    exp = bytearray(struct.pack("<IBBBBHxx", 0x42504600, 1, 0, 0, 16, 0))
    for i, rd in enumerate(reads):
        rid = b"r%d " % i
        exp += struct.pack("<H", len(rid)) + rid + struct.pack("<Q", len(rd)) + pml[int(offs[i]): int(offs[i + 1])].astype("<u2").tobytes()
    assert open(prefix + ".pml.bpf", "rb").read() == bytes(exp)
    cpu.close()


def test_ahead_rows_flag(movi_bin, oracles, tmp_path):
    """`movi query --ahead-rows 0|1` (extension): the look-ahead rows off or built on request -- the same BPF bytes and count
    lines either way."""
    from oracle import build_index as B
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    reads_path = str(tmp_path / "mixed.fa")
    write_mixed_reads(reads_path, np.random.default_rng(99), ref)
    outs, counts = [], []
    for v in ("0", "1"):
        prefix = str(tmp_path / ("o" + v))
        r = run(["query", "-i", IDX[6], "-r", reads_path, "-o", prefix, "--ahead-rows", v, "-n"])
        assert r.returncode == 0, r.stderr
        outs.append(open(prefix + ".pml.bpf", "rb").read())
        r = run(["query", "-i", IDX[6], "-r", reads_path, "--count", "--stdout", "--ahead-rows", v, "-n"])
        assert r.returncode == 0, r.stderr
        counts.append(r.stdout)
    assert outs[0] == outs[1] and len(outs[0]) > 100000
    assert counts[0] == counts[1] and counts[0].count(b"\n") > 100
