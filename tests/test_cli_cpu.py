"""CPU suite for the C++ host CLI: argument handling, read parsing / id rule, the
reference's batch cut and strand-order emulation (checked against an independent
round-by-round simulation written from src/batch_loader.cpp + src/read_processor.cpp),
and `movi view` on hand-built BPF bytes.  No GPU work."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT

MOVI = os.path.join(ROOT, "movi_amd", "bin", "movi")


@pytest.fixture(scope="module")
def movi_bin(built_lib):
    if not os.path.exists(MOVI):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "movi_amd", "csrc")], stdout=subprocess.DEVNULL)
    return MOVI


def run(args, **kw):
    return subprocess.run([MOVI] + args, capture_output=True, **kw)


def test_argument_errors(movi_bin):
    # set_count / set_zml / set_pml are applied in this order and clear each other (movi_parser.cpp:353-355,
    # movi_options.hpp:108-110): --pml --count is a PML query, not a usage error
    r = run(["query", "-i", "x", "-r", "y", "--pml", "--count", "--zml"])
    assert r.returncode == 1 and b"Error parsing command line" not in r.stderr
    r = run(["query", "-i", "x"])
    assert r.returncode == 1 and b"Please include one index directory and one read file." in r.stderr
    r = run(["query", "-i", "x", "-r", "y", "--mem"])
    assert r.returncode == 1 and b"not supported" in r.stderr
    r = run(["null"])
    assert r.returncode == 1 and b"Please specify the index directory file." in r.stderr       # movi_parser.cpp:538-540
    r = run(["null", "-i", "x", "--gen-reads"])
    assert r.returncode == 1 and b"Please specify the reference fasta file." in r.stderr       # movi_parser.cpp:531-533
    r = run(["build", "-i", "x", "-f", "y"])
    assert r.returncode == 1
    r = run(["query", "-i", "x", "-r", "y", "--ignore-illegal-chars", "3"])
    assert r.returncode == 1 and b"ignore-illegal-chars should be either 1" in r.stderr
    r = run(["view", "--bpf", "/nonexistent.bpf"])
    assert r.returncode == 1 and b"Failed to open the MLS file" in r.stderr
    # engine extensions and flags that are accepted and implied: parsed before any device is touched
    r = run(["query", "-i", "x", "-r", "y", "--seg-len", "abc"])
    assert r.returncode == 1 and b"failed to parse for option 'seg-len'" in r.stderr
    r = run(["query", "-i", "x", "-r", "y", "--ahead-rows", "3"])
    assert r.returncode == 1 and b"--ahead-rows must be 0 or 1" in r.stderr
    r = run(["query", "-i", "x", "-r", "y", "--mmap", "--seg-len", "64"])
    assert r.returncode == 1 and b"not supported" not in r.stderr and b"Error parsing command line" not in r.stderr


def reference_schedule(lines, fmt, strands, prefetch):
    """Independent restatement: loadBatch (batch_loader.cpp:26-89), grabNextRead (:91-143),
    process_latency_hiding (read_processor.cpp:641-730) simulated one round at a time."""
    min_reads = 4 * strands if prefetch else 1
    out, pos, batch_no = [], 0, 0
    while pos < len(lines):
        bases = reads = nl = rec = 0
        batch = []
        while pos < len(lines) and (bases < 1000 or reads < min_reads):
            ln = lines[pos]; pos += 1; nl += 1; rec += len(ln)
            if fmt == "fq":
                if nl % 4 == 0:
                    bases += rec // 2; rec = 0; reads += 1
            elif pos < len(lines) and lines[pos][:1] == b">":
                bases += rec; rec = 0; reads += 1
            batch.append(ln)
        # parse the batch
        recs, p = [], 0
        while p < len(batch):
            h = batch[p]
            if not h:
                break
            k = next((i for i in range(1, len(h)) if h[i:i + 1] in (b" ", b"\t", b"\r")), len(h))
            rid = h[1:1 + k]
            p += 1
            if fmt == "fq":
                seq = batch[p].rstrip(); p += 3
            else:
                seq = b""
                while p < len(batch) and batch[p][:1] != b">":
                    seq += batch[p].rstrip(); p += 1
            recs.append((rid, len(seq)))
        if not prefetch:
            out += [(batch_no, r, l) for r, l in recs]
        else:
            nxt = 0
            cur = [None] * strands
            left = [0] * strands
            for s in range(strands):
                if nxt < len(recs):
                    cur[s], left[s] = recs[nxt], max(recs[nxt][1], 1); nxt += 1
            while any(c is not None for c in cur):
                for s in range(strands):
                    if cur[s] is None:
                        continue
                    left[s] -= 1
                    if left[s] == 0:
                        out.append((batch_no, cur[s][0], cur[s][1]))
                        if nxt < len(recs):
                            cur[s], left[s] = recs[nxt], max(recs[nxt][1], 1); nxt += 1
                        else:
                            cur[s] = None
        batch_no += 1
    return out


def make_reads(rng, n, fmt):
    lines = []
    for i in range(n):
        L = int(rng.integers(1, 400)) if rng.random() < 0.9 else int(rng.integers(400, 3000))
        seq = bytes(rng.choice(list(b"ACGTN"), size=L).astype(np.uint8))
        cm = [b"", b" comment here", b"\tx=1", b" "][int(rng.integers(0, 4))]
        if fmt == "fq":
            lines += [b"@q%d" % i + cm, seq, b"+", b"I" * L]
        else:
            lines.append(b">r%d" % i + cm)
            w = int(rng.integers(40, 120))
            lines += [seq[j:j + w] for j in range(0, L, w)]
    return lines


@pytest.mark.parametrize("fmt", ["fa", "fq"])
@pytest.mark.parametrize("flags,strands,prefetch", [(["-s16"], 16, True), (["-s4"], 4, True), (["-s", "1"], 1, True),
                                                    (["-n"], 16, False)])
def test_batching_and_record_order(movi_bin, tmp_path, fmt, flags, strands, prefetch):
    rng = np.random.default_rng(len(flags) * 7 + strands + (fmt == "fq"))
    lines = make_reads(rng, 600, fmt)
    path = tmp_path / ("reads." + fmt)
    path.write_bytes(b"\n".join(lines) + b"\n")
    r = run(["plan", "-r", str(path)] + flags)
    assert r.returncode == 0, r.stderr
    got = []
    for l in r.stdout.split(b"\n"):
        if l:
            b, rest = l.split(b"\t", 1)          # ids may themselves end in a tab
            i, ln = rest.rsplit(b"\t", 1)
            got.append((b, i, ln))
    exp = reference_schedule(lines, fmt, strands, prefetch)
    assert len(got) == len(exp) == 600
    assert [(int(b), i, int(l)) for b, i, l in got] == exp
    # the id keeps the first whitespace character after the name (batch_loader.cpp:117-119)
    ids = [i for _, i, _ in got]
    assert any(i.endswith(b" ") for i in ids) and any(i.endswith(b"\t") for i in ids)


@pytest.mark.parametrize("fmt", ["fa", "fq"])
def test_parallel_mmap_parser_equals_stream_parser(movi_bin, tmp_path, fmt):
    """A read file big enough for the parser's worker threads (> 4 MB of sequence), with comment headers, multi-line
    FASTA records, CRLF line ends and a last line without a newline: the memory-mapped parallel path, the stream path
    (MOVI_NO_MMAP=1) and stdin give the same batches, ids, lengths and order -- and the reference's schedule."""
    rng = np.random.default_rng(77 + (fmt == "fq"))
    lines = make_reads(rng, 14000, fmt)
    for k in range(0, len(lines), 97):                        # sprinkle CRLF line ends (stripped by the reader, kept in ids)
        if not lines[k].startswith((b">", b"@", b"+")):
            lines[k] = lines[k] + b"\r"
    path = tmp_path / ("big." + fmt)
    path.write_bytes(b"\n".join(lines))                       # no trailing newline
    assert path.stat().st_size > (4 << 20)
    a = run(["plan", "-r", str(path), "-s16"])
    b = run(["plan", "-r", str(path), "-s16"], env=dict(os.environ, MOVI_NO_MMAP="1"))
    c = run(["plan", "-r", "-", "-s16"], input=path.read_bytes())
    assert a.returncode == b.returncode == c.returncode == 0, (a.stderr, b.stderr, c.stderr)
    assert a.stdout == b.stdout == c.stdout and a.stdout.count(b"\n") == 14000
    # the same file cut into many small chunks: the parallel newline scan then covers ~1 MB windows, so chunks, reference
    # batches and lines straddle its windows (lines beyond a window are found the slow way) -- same plan, worker count or not
    # (MOVI_PLAN_THREADS: the strand-scheduler emulation by ranges of batches on a pool, as the writer stage of `movi query` runs it)
    for env in (dict(MOVI_CHUNK_BASES="50000"), dict(MOVI_CHUNK_BASES="50000", MOVI_NO_AFFINITY="1"), dict(MOVI_CHUNK_BASES="7"),
                dict(MOVI_PLAN_THREADS="5"), dict(MOVI_PLAN_THREADS="3", MOVI_CHUNK_BASES="500000")):
        d = run(["plan", "-r", str(path), "-s16"], env=dict(os.environ, **env))
        assert d.returncode == 0 and d.stdout == a.stdout, env
    clean = [l[:-1] if l.endswith(b"\r") and not l.startswith((b">", b"@")) else l for l in lines]
    exp = reference_schedule(clean, fmt, 16, True)
    got = []
    for l in a.stdout.split(b"\n"):
        if l:
            bb, rest = l.split(b"\t", 1)
            i, ln = rest.rsplit(b"\t", 1)
            got.append((int(bb), i, int(ln)))
    assert [(x, z) for x, _, z in got] == [(x, z) for x, _, z in exp]     # batches and lengths in the reference's order
    assert [y for _, y, _ in got] == [y for _, y, _ in exp]


def test_view_prints_read_order(movi_bin, tmp_path):
    """BPF layout: 12-byte header | per read u16 id_len, id, u64 n, n x u16 (last base first);
    `view` prints each record reversed (src/movi.cpp:454-456)."""
    recs = [(b"readA ", [0, 1, 2, 65535, 7]), (b"b", []), (b"c\t", [9])]
    blob = struct.pack("<IBBBBHxx", 0x42504600, 1, 0, 0, 16, 0)
    for rid, vals in recs:
        blob += struct.pack("<H", len(rid)) + rid + struct.pack("<Q", len(vals)) + np.asarray(vals, "<u2").tobytes()
    f = tmp_path / "x.bpf"
    f.write_bytes(blob)
    r = run(["view", "--bpf", str(f)])
    assert r.returncode == 0
    exp = b"".join(b">" + rid + b"\n" + b"".join(b"%d " % v for v in vals[::-1]) + b"\n" for rid, vals in recs)
    assert r.stdout == exp
    bad = tmp_path / "bad.bpf"
    bad.write_bytes(b"\x00" * 12)
    assert run(["view", "--bpf", str(bad)]).returncode == 1


@pytest.mark.parametrize("entry_bits", [16, 32, 64])
def test_view_large_file_parallel_path(movi_bin, tmp_path, entry_bits):
    """A BPF file big enough for `view`'s worker threads (several batches of 32 M values would be too slow to check in
    Python: 3 M values here, which still crosses the 1 M-value threshold of the parallel path), ragged record sizes, an
    empty record, a name with an embedded NUL, every entry size of the reference (16 / 32 / 64 bits); and a truncated
    copy, which must fall back to the record-by-record reader and print the complete records."""
    rng = np.random.default_rng(entry_bits)
    dt = {16: "<u2", 32: "<u4", 64: "<u8"}[entry_bits]
    blob = [struct.pack("<IBBBBHxx", 0x42504600, 1, 0, 0, entry_bits, 0)]
    exp = []
    for i in range(1500):
        n = int(rng.integers(0, 4000)) if i != 7 else 0
        vals = rng.integers(0, 70000 if entry_bits > 16 else 65536, n).astype(dt)
        rid = b"r%d " % i if i != 11 else b"nul\x00tail"
        blob.append(struct.pack("<H", len(rid)) + rid + struct.pack("<Q", n) + vals.tobytes())
        exp.append(b">" + rid.split(b"\x00")[0] + b"\n" + b"".join(b"%d " % v for v in vals[::-1].tolist()) + b"\n")
    f = tmp_path / "big.bpf"
    f.write_bytes(b"".join(blob))
    r = run(["view", "--bpf", str(f)])
    assert r.returncode == 0 and r.stdout == b"".join(exp)
    cut = tmp_path / "cut.bpf"
    cut.write_bytes(b"".join(blob)[:-5])
    r2 = run(["view", "--bpf", str(cut)])
    assert r2.returncode == 0 and r2.stdout.startswith(b"".join(exp[:-1]))


@pytest.mark.parametrize("type_name,mode,separators,size", [
    ("regular-thresholds", 6, False, 948119), ("blocked-thresholds", 8, False, 711733), ("sampled-thresholds", 7, False, 475326),
    ("regular", 3, False, 871479), ("blocked", 2, False, 654253), ("sampled", 5, False, 437006),
    ("regular-thresholds", 6, True, 948232), ("blocked", 2, True, 654280)])
def test_movi_build_reproduces_reference_index_sizes(movi_bin, tmp_path, type_name, mode, separators, size):
    """`movi build -i DIR -f ref.fasta --type T [--separators]`: the in-memory constructor behind the host CLI reproduces the
    reference's index-size known answers (tests/test_build.cpp:33-95) and the numpy constructor byte for byte."""
    from conftest import GOLDEN
    from oracle import build_index as B
    d = tmp_path / "ix"
    r = run(["build", "-i", str(d), "-f", os.path.join(GOLDEN, "ref.fasta"), "--type", type_name] + (["--separators"] if separators else []))
    assert r.returncode == 0, r.stderr
    img = (d / "index.movi").read_bytes()
    assert len(img) == size
    ref = B.read_fasta(os.path.join(GOLDEN, "ref.fasta"))[0][1]
    assert img == B.build_index_from_seqs([ref], mode, separators=separators)


def test_movi_build_argument_errors(movi_bin, tmp_path):
    from conftest import GOLDEN
    r = run(["build", "-f", os.path.join(GOLDEN, "ref.fasta")])
    assert r.returncode == 1 and b"Please specify the index directory file." in r.stderr
    r = run(["build", "-i", str(tmp_path / "x")])
    assert r.returncode == 1 and b"Please specify the reference fasta file." in r.stderr
    r = run(["build", "-i", str(tmp_path / "x"), "-f", os.path.join(GOLDEN, "ref.fasta"), "--type", "constant"])
    assert r.returncode == 1 and b"not supported" in r.stderr
    r = run(["build", "-i", str(tmp_path / "x"), "-f", str(tmp_path / "missing.fa")])
    assert r.returncode == 1 and b"cannot open" in r.stderr


@pytest.mark.parametrize("content,message", [
    (b">ab\nACGT\n>b\nAC\n", b"header line is missing an id"),                 # a 2-character header (batch_loader.cpp:108-110)
    (b"ACGT\n", b"unrecognized input query file type"),
    (b">\nACGT\n", b"header line is missing an id"),
    (b"@q1\nACGT\n+\nIIII\nq2x\nAC\n+\nII\n", b"Incorrect FASTQ entry"),
    (b">r1\nACGT\n>r2\nAC\n" * 3 + b"@r3\nAC\n", None)])                     # '@' line inside FASTA = sequence text, as in the reference
def test_malformed_inputs_fail_the_same_way_on_both_parser_paths(movi_bin, tmp_path, content, message):
    path = tmp_path / "bad.fx"
    path.write_bytes(content)
    a = run(["plan", "-r", str(path), "-n"])
    b = run(["plan", "-r", str(path), "-n"], env=dict(os.environ, MOVI_NO_MMAP="1"))
    assert (a.returncode, a.stdout, a.stderr) == (b.returncode, b.stdout, b.stderr)
    if message is None:
        assert a.returncode == 0
    else:
        assert a.returncode == 1 and message in a.stderr
    empty = tmp_path / "empty.fa"
    empty.write_bytes(b"")
    assert run(["plan", "-r", str(empty)]).returncode == 0


def test_lines_longer_than_the_newline_scan_window(movi_bin, tmp_path):
    """A FASTA whose sequence lines are longer than the parser's scan-ahead window (2.5 MB single-line records between short
    ones, chunks of 1000 bases): ids, lengths and order as the stream parser and as written."""
    rng = np.random.default_rng(5)
    lens = [10, 2_500_000, 33, 1_200_000, 1, 70, 3_000_001, 5]
    recs = [(b"x%d" % i, bytes(rng.choice(list(b"ACGT"), size=L).astype(np.uint8))) for i, L in enumerate(lens)]
    path = tmp_path / "long_lines.fa"
    path.write_bytes(b"".join(b">" + i + b" c\n" + s + b"\n" for i, s in recs))
    a = run(["plan", "-r", str(path), "-n"], env=dict(os.environ, MOVI_CHUNK_BASES="1000"))
    b = run(["plan", "-r", str(path), "-n"], env=dict(os.environ, MOVI_NO_MMAP="1"))
    assert a.returncode == 0 and a.stdout == b.stdout
    got = [(l.split(b"\t")[1], int(l.rsplit(b"\t", 1)[1])) for l in a.stdout.split(b"\n") if l]
    assert got == [(i + b" ", len(s)) for i, s in recs]


def _plan3(path, flags, extra_env=None):
    """`movi plan` through the three cuts: scanned lines in bulk (default on a mapped file), line by line on the mapped file,
    line by line on a stream."""
    env = dict(os.environ, **(extra_env or {}))
    a = run(["plan", "-r", str(path)] + flags, env=env)
    b = run(["plan", "-r", str(path)] + flags, env=dict(env, MOVI_NO_FAST_CUT="1"))
    c = run(["plan", "-r", str(path)] + flags, env=dict(env, MOVI_NO_MMAP="1"))
    return a, b, c


@pytest.mark.parametrize("fmt", ["fa", "fa1", "fq"])
@pytest.mark.parametrize("flags", [["-s16"], ["-s", "1"], ["-n"]])
def test_bulk_batch_cut_equals_the_line_by_line_cut(movi_bin, tmp_path, fmt, flags):
    """The parser cuts the lines its newline scan has found into reads and reference batches in bulk (BatchReader::cut_ahead)
    where the input is regular, and line by line (loadBatch / grabNextRead as the reference runs them) everywhere else:
    same batches, ids, lengths and order on well-formed files of every shape, whatever the chunk size -- and on files with
    an irregular line somewhere in the middle, where the bulk cut must stand back."""
    rng = np.random.default_rng(400 + len(flags[0]) + len(fmt))
    n = 30000
    if fmt == "fa1":                                          # single-line FASTA, short reads: the common case
        lines = []
        for i in range(n):
            lines += [b">read%d" % i, bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(30, 160))).astype(np.uint8))]
    else:
        lines = make_reads(rng, n, fmt)
    variants = {
        "regular": b"\n".join(lines) + b"\n",
        "no_final_newline": b"\n".join(lines),
        "blank_tail": b"\n".join(lines) + b"\n\n\n",
        "blank_line_inside": b"\n".join(lines[: len(lines) // 2 // 4 * 4] + [b""] + lines[len(lines) // 2 // 4 * 4:]) + b"\n",
        "crlf": b"\r\n".join(lines) + b"\r\n",
        # several irregular lines: the bulk cut takes the records before each, stands back across it, and resumes behind it
        "blank_lines_every_few_thousand": b"\n".join(l + b"\n" if i % 9000 == 8999 and (fmt != "fq" or i % 4 == 3) else l
                                                     for i, l in enumerate(lines)) + b"\n",
    }
    for name, content in variants.items():
        path = tmp_path / ("%s.%s" % (name, fmt))
        path.write_bytes(content)
        for env in ({}, {"MOVI_CHUNK_BASES": "400000"}, {"MOVI_CHUNK_BASES": "400000", "MOVI_NO_AFFINITY": "1"}):
            a, b, c = _plan3(path, flags, env)
            assert (a.returncode, a.stdout, a.stderr) == (b.returncode, b.stdout, b.stderr) == (c.returncode, c.stdout, c.stderr), (name, env)
        if name == "regular":
            assert a.returncode == 0 and a.stdout.count(b"\n") == n


@pytest.mark.parametrize("fmt,bad,message", [("fa1", b">a", b"header line is missing an id"),
                                             ("fq", b"q_without_at", b"Incorrect FASTQ entry"),
                                             ("fq", b"@", b"header line is missing an id")])
def test_bulk_batch_cut_leaves_malformed_records_to_the_reference_path(movi_bin, tmp_path, fmt, bad, message):
    rng = np.random.default_rng(500 + len(bad))
    lines = []
    for i in range(20000):
        seq = bytes(rng.choice(list(b"ACGT"), size=100).astype(np.uint8))
        lines += [b"@r%d" % i, seq, b"+", b"I" * 100] if fmt == "fq" else [b">r%d" % i, seq]
    k = (len(lines) // 2) // 4 * 4
    lines[k] = bad                                            # a header in the middle of the file
    path = tmp_path / ("bad." + fmt)
    path.write_bytes(b"\n".join(lines) + b"\n")
    a, b, c = _plan3(path, ["-s16"])
    assert (a.returncode, a.stdout, a.stderr) == (b.returncode, b.stdout, b.stderr) == (c.returncode, c.stdout, c.stderr)
    assert a.returncode == 1 and message in a.stderr


@pytest.mark.parametrize("fmt", ["fa", "fq"])
@pytest.mark.parametrize("strands", [16, 4, 1])
def test_closed_form_batch_cut_equals_the_running_sum(movi_bin, tmp_path, fmt, strands):
    """Round 5: where every record in reach counts for at least ceil(1000 / min_reads) "bases", loadBatch's rule ends every batch
    after exactly min_reads records and the bulk cut says so in closed form; a record too short for that sends the pass back to
    the running sum.  Three files -- regular 150 bp reads, tiny reads (1 - 8 bases, one-letter ids: far below the bound), and
    stretches of both -- must give the plan of the running sum (MOVI_NO_CLOSED_CUT=1) and of the line-by-line cut
    (MOVI_NO_FAST_CUT=1), chunked or not."""
    rng = np.random.default_rng(50 + strands + (fmt == "fq"))

    def rec(i, L, short_id):
        s = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), L))
        name = (b"%c" % (97 + i % 26)) * 2 if short_id else b"read%d" % i
        return [b"@" + name, s, b"+", b"I" * L] if fmt == "fq" else [b">" + name, s]

    files = {"regular": [l for i in range(12000) for l in rec(i, 150, False)],
             "tiny": [l for i in range(30000) for l in rec(i, int(rng.integers(1, 9)), True)],
             "mixed": [l for i in range(30000) for l in rec(i, 150 if (i // 3000) % 2 == 0 else int(rng.integers(1, 9)), (i // 3000) % 2 == 1)]}
    for name, lines in files.items():
        path = tmp_path / ("%s.%s" % (name, fmt))
        path.write_bytes(b"\n".join(lines) + b"\n")
        flags = ["plan", "-r", str(path), "-s%d" % strands]
        a = run(flags)
        b = run(flags, env=dict(os.environ, MOVI_NO_CLOSED_CUT="1"))
        c = run(flags, env=dict(os.environ, MOVI_NO_FAST_CUT="1"))
        d = run(flags, env=dict(os.environ, MOVI_CHUNK_BASES="200000"))
        assert a.returncode == b.returncode == c.returncode == d.returncode == 0, (name, a.stderr, b.stderr, c.stderr, d.stderr)
        assert a.stdout == b.stdout == c.stdout == d.stdout and a.stdout.count(b"\n") == len(lines) // (4 if fmt == "fq" else 2), name
