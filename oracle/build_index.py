"""build_index.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy restatement of the reference's offline index construction for modes 6
(regular-thresholds) and 8 (blocked-thresholds): FASTA -> cleaned text with
reverse complements -> BWT + thresholds -> move rows -> `index.movi` bytes.

It exists so that the oracle (oracle/movi_oracle.c) and the HIP path can be
pinned against the golden vectors the reference's own tests hold:
`tests_data/sample.fastq.pmls.sorted` is defined on an index of
`tests_data/ref.fasta` (tests/test_pml.cpp:6-55), and tests/test_build.cpp:37,53
assert that index's size (948119 B mode 6, 711733 B mode 8).  Reproducing both
validates every convention below.

Reference code restated (file:line relative to /root/reference):
  * src/prepare_ref.cpp:16-85        FASTA cleaning + reverse complement
  * external pfp-thresholds (tag `movi`, fetched by CMakeLists.txt:80-114, NOT in
    the checkout): produces ref.bwt (raw BWT, terminator byte 0) and ref.thr_pos
    (one 5-byte LE position per BWT run).  Its published definition (Bannai,
    Gagie, I 2020; Rossi et al. 2022 "MONI"): the threshold of a run of c is the
    position of the minimum LCP between the end of the previous run of c and the
    start of this run; leftmost minimum, 0 when there is no previous run.
  * src/move_structure_build.cpp:17-72    build()
  * :223-426 detect_move_row_boundaries (non-preprocessed branch :328-396)
  * :74-121  find_run_heads_information, :449-692 build_move_rows
  * :694-731 find_base_interval_data,   :807-935 compute_thresholds
  * :939-1074 compute_blocked_ids
  * src/move_row.cpp:118-177 (mode 6 setters), :181-266 (mode 8 setters)
  * src/move_structure_io.cpp:435-469 serialize
"""
import numpy as np

MOVI_MAGIC = 0x4D4F5649          # include/utils.hpp:29
MAX_RUN = {6: 2047, 8: 1023, 7: 511, 5: 1023, 3: 4095, 2: 1023}   # include/move_row_configs.hpp:51,101,135,117,31,72
THRESHOLD_MODES = (6, 7, 8)      # include/utils.hpp:146 (USE_THRESHOLDS); mode 5 "sampled" keeps none
TALLY_CHECKPOINTS = 20           # include/movi_options.hpp:257 (movi build --checkpoint default)
BLOCK_SIZE = {8: 1 << 20, 2: 1 << 22}                      # include/move_row_configs.hpp:102 / :73
MAX_ALLOWED_BLOCKED_ID = {8: (1 << 22) - 1, 2: (1 << 24) - 1}   # :103 / :74

ALPHAMAP_3 = np.array([[3, 0, 1, 2], [0, 3, 1, 2], [0, 1, 3, 2], [0, 1, 2, 3]])  # src/utils.cpp:5-8


def read_fasta(path):
    """kseq-style: returns list of (name, sequence bytes)."""
    recs, name, chunks = [], None, []
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if name is not None:
                    recs.append((name, b"".join(chunks)))
                name, chunks = line[1:].split()[0] if line[1:].split() else b"", []
            elif name is not None:
                chunks.append(line.strip())
    if name is not None:
        recs.append((name, b"".join(chunks)))
    return recs


SEPARATOR = 37                   # '%', include/commons.hpp:63


def clean_text(seqs, rc=True, separators=False):
    """src/prepare_ref.cpp:39-58: anything that is not an upper-case A/C/G/T in
    the input (lower case included -- the test uses the pre-uppercasing byte)
    becomes 'A'; each record is followed by its reverse complement.  With
    `separators` (movi build --separators, prepare_ref.cpp:61-66) every record and
    every reverse complement is followed by one '%'.
    Returns the text as uint8 with one trailing 0 terminator."""
    comp = np.zeros(256, np.uint8)
    comp[ord("A")], comp[ord("C")], comp[ord("G")], comp[ord("T")] = ord("T"), ord("G"), ord("C"), ord("A")
    parts = []
    for s in seqs:
        a = np.frombuffer(bytes(s), np.uint8).copy()
        ok = (a == 65) | (a == 67) | (a == 71) | (a == 84)
        a[~ok] = 65
        parts.append(a)
        if separators:
            parts.append(np.full(1, SEPARATOR, np.uint8))
        if rc:
            parts.append(comp[a[::-1]])
            if separators:
                parts.append(np.full(1, SEPARATOR, np.uint8))
    parts.append(np.zeros(1, np.uint8))
    return np.concatenate(parts)


def suffix_array(t):
    """Prefix doubling; t ends with a unique smallest terminator."""
    n = len(t)
    rank = t.astype(np.int64)
    k = 1
    while True:
        r2 = np.zeros(n, np.int64)
        r2[: n - k] = rank[k:] + 1
        key = rank * (n + 2) + r2
        sa = np.argsort(key, kind="stable")
        ks = key[sa]
        newrank = np.empty(n, np.int64)
        newrank[sa] = np.concatenate(([0], np.cumsum(ks[1:] != ks[:-1])))
        rank = newrank
        if rank[sa[-1]] == n - 1:
            return sa
        k *= 2


def lcp_array(t, sa):
    """LCP[i] = lcp(suffix sa[i-1], suffix sa[i]); LCP[0] = 0.  Vectorised
    doubling comparison (numpy), exact."""
    n = len(t)
    rank = np.empty(n, np.int64)
    rank[sa] = np.arange(n)
    # Kasai, but batched: process in text order with the h-1 lower bound is
    # inherently sequential; n here is small (<= a few Mbp) so do it in pure
    # python over memoryviews.
    tb = t.tobytes()
    lcp = np.zeros(n, np.int64)
    sal = sa.tolist()
    rk = rank.tolist()
    h = 0
    out = [0] * n
    for i in range(n):
        r = rk[i]
        if r > 0:
            j = sal[r - 1]
            while i + h < n and j + h < n and tb[i + h] == tb[j + h]:
                h += 1
            out[r] = h
            if h > 0:
                h -= 1
        else:
            h = 0
    lcp[:] = out
    return lcp


def bwt_and_thresholds(t):
    """BWT of t (t[-1] == 0 is the terminator) and the per-run threshold
    positions in pfp-thresholds' .thr_pos convention (see module docstring)."""
    sa = suffix_array(t)
    bwt = t[sa - 1]                     # sa==0 -> t[-1] == terminator
    lcp = lcp_array(t, sa)
    n = len(bwt)
    starts = np.flatnonzero(np.concatenate(([True], bwt[1:] != bwt[:-1])))
    ends = np.concatenate((starts[1:], [n])) - 1
    heads = bwt[starts]
    thr = np.zeros(len(starts), np.int64)
    last_end = {}
    st, en, hd = starts.tolist(), ends.tolist(), heads.tolist()
    for k in range(len(st)):
        c = hd[k]
        if c in last_end:
            e = last_end[c]
            thr[k] = e + 1 + int(np.argmin(lcp[e + 1: st[k] + 1]))
        last_end[c] = en[k]
    return bwt, thr


def build_rows(bwt, thr, mode):
    """Everything MoveStructure::build() derives from ref.bwt + ref.thr_pos.
    Returns a dict of the fields serialize() writes."""
    assert mode in (2, 3, 5, 6, 7, 8)
    n = len(bwt)
    maxrun = MAX_RUN[mode]
    # --- detect_move_row_boundaries (:328-396) + fill_bits_by_thresholds (:733-746)
    hard = np.zeros(n + 1, bool)
    hard[0] = True
    hard[1:n] = bwt[1:] != bwt[:-1]
    orig_starts = np.flatnonzero(hard[:n])
    original_r = len(orig_starts)
    assert len(thr) == original_r
    if mode in THRESHOLD_MODES:
        hard[thr] = True                                # bits[thresholds[i]] = 1 (rows split at thresholds, :733-746)
    seg = np.flatnonzero(hard[:n])
    seg_len = np.diff(np.concatenate((seg, [n])))
    # split every segment into pieces of MAX_RUN_LENGTH (:380-385: a new row starts
    # once the current one holds MAX_RUN_LENGTH characters)
    pieces = (seg_len + maxrun - 1) // maxrun
    row_seg = np.repeat(np.arange(len(seg)), pieces)
    first_of_seg = np.concatenate(([0], np.cumsum(pieces)[:-1]))
    within = np.arange(len(row_seg)) - first_of_seg[row_seg]
    all_p = seg[row_seg] + within * maxrun
    r = len(all_p)
    lens = np.diff(np.concatenate((all_p, [n])))
    heads = bwt[all_p]
    # --- build_alphabet (:428-447)
    alphamap = np.full(256, 256, np.uint64)
    alphabet, counts = [], []
    for ch in range(1, 256):
        cnt = int(np.count_nonzero(bwt == ch))
        if cnt:
            alphamap[ch] = len(alphabet)
            alphabet.append(ch)
            counts.append(cnt)
    sigma = len(alphabet)
    # MoveStructure::use_separator, src/move_structure.cpp:547-552: a 5-symbol alphabet led by '%'
    sep = 1 if (sigma == 5 and alphabet[0] == SEPARATOR) else 0
    assert 1 <= sigma <= 4 or sep, "only DNA alphabets (<= 4 symbols, or '%' + 4 with separators) are in scope"
    code = np.where(heads == 0, 0, alphamap[heads].astype(np.int64)).astype(np.int64)
    code[heads == 0] = 0                                # set_c: alphamap[0]==256 shifts out
    end_bwt_idx = int(np.flatnonzero(heads == 0)[0])
    is_end = np.arange(r) == end_bwt_idx
    # --- find_run_heads_information (:74-121) + LF_heads (src/move_structure.cpp:515-523)
    C = 1 + np.concatenate(([0], np.cumsum(counts)[:-1]))
    lf = np.zeros(r, np.int64)
    for a in range(sigma):
        m = (code == a) & ~is_end
        cl = np.where(m, lens, 0)
        heads_rank = np.cumsum(cl) - cl
        lf[m] = C[a] + heads_rank[m]
    lf[is_end] = 0
    # --- build_move_rows (:449-692)
    pp_id = np.searchsorted(all_p, lf, side="right") - 1
    offset = lf - all_p[pp_id]
    assert offset.max() <= maxrun and lens.max() <= maxrun
    # --- find_base_interval_data (:694-731)
    first_runs, first_offsets, last_runs, last_offsets = [0], [0], [0], [0]
    char_count = 1
    for a in range(sigma):
        lr, lo = last_runs[-1], last_offsets[-1]
        if lo + 1 >= lens[lr]:
            first_runs.append(lr + 1); first_offsets.append(0)
        else:
            first_runs.append(lr); first_offsets.append(lo + 1)
        char_count += counts[a]
        occ_rank = int(np.searchsorted(all_p, char_count, side="left"))   # rbits(char_count)
        last_runs.append(occ_rank - 1)
        last_offsets.append(char_count - int(all_p[occ_rank - 1]) - 1)
    # --- compute_thresholds (:807-935), split mode: one bit per (row, other DNA char).  With separators
    # (:826-831, :836-858, :912-921) no threshold is kept FOR the separator; a row OF the separator -- and
    # the '$' row, whose character field decodes as the separator -- gets an explicit 4-value entry
    # in separators_thresholds (appended in descending row order, row 0 last), keyed by row in the map.
    thr_bits = np.zeros((r, 3), np.int64)
    end_thr = [0, 0, 0, 0]
    sep_thr, sep_map = [], {}
    alphabet_thresholds = [n] * sigma
    thr_i = original_r - 1
    cl, pl, ll, tl = code.tolist(), all_p.tolist(), lens.tolist(), thr.tolist()
    for i in range(r - 1, 0, -1) if mode in THRESHOLD_MODES else ():
        rc = cl[i]                                      # '$' row has c == 0 -> 'A' (:823), '%' with separators
        if sep and rc == 0:
            sep_thr.append([0, 0, 0, 0])
            sep_map[i] = len(sep_thr) - 1
        for j in range(sigma):
            if j == rc:
                alphabet_thresholds[j] = tl[thr_i]
            else:
                if sep and j == 0:
                    continue                            # :849-852
                cur = alphabet_thresholds[j]
                if cur >= pl[i] + ll[i]:
                    val, bit = ll[i], 1
                elif cur <= pl[i]:
                    val, bit = 0, 0
                else:
                    val, bit = cur - pl[i], None        # strictly inside the row (:869-871)
                if i == end_bwt_idx:
                    end_thr[j - sep] = val              # set_threshold_for_one_character :776-779
                elif sep and rc == 0:
                    sep_thr[-1][j - 1] = val            # :781-784
                else:
                    if bit is None:
                        raise AssertionError("threshold strictly inside a row: rows must be split at thresholds")
                    thr_bits[i, ALPHAMAP_3[rc - sep][j - sep]] = bit
        if cl[i] != cl[i - 1] or i == end_bwt_idx or i - 1 == end_bwt_idx:
            thr_i -= 1
    if mode not in THRESHOLD_MODES:
        pass                                            # no thresholds of any kind (write_separators_thresholds is under USE_THRESHOLDS too)
    elif sep and cl[0] == 0:                            # :917-920
        sep_thr.append([0, 0, 0, 0])
        sep_map[0] = len(sep_thr) - 1
    else:
        thr_bits[0, :] = 0                              # :903-911, :922-929
    out = dict(mode=mode, n=n, r=r, original_r=original_r, end_bwt_idx=end_bwt_idx,
               alphamap=alphamap, alphabet=bytes(alphabet), counts=counts,
               first_runs=first_runs, first_offsets=first_offsets,
               last_runs=last_runs, last_offsets=last_offsets, end_thr=end_thr,
               lens=lens, offset=offset, code=code, pp_id=pp_id, thr_bits=thr_bits, all_p=all_p,
               sep=sep, sep_thr=sep_thr, sep_map=sep_map)
    if mode in (8, 2):
        out.update(compute_blocked_ids(pp_id, code, end_bwt_idx, first_runs, sigma, mode))
    if mode in (5, 7):
        out.update(compute_tally_ids(pp_id, code, end_bwt_idx, sigma))
    return out


def compute_tally_ids(pp_id, code, end_bwt_idx, sigma, checkpoints=TALLY_CHECKPOINTS):
    """Sampled ("tally") modes keep no id in the row: src/move_structure_build.cpp:486-496, :571-596, :677-682.
    Every `checkpoints` rows the destination id of the latest run of EACH character seen so far is stored
    (for the checkpoint row's own character that is the row itself); a character not seen yet gets the id
    of its first run once that shows up; one extra last entry holds the final ids."""
    r = len(pp_id)
    n_ck = r // checkpoints + 2
    tally = np.zeros((sigma, n_ck), np.int64)
    cur = [r] * sigma
    ids, cl = pp_id.tolist(), code.tolist()
    for i in range(r):
        if i != end_bwt_idx:
            a = cl[i]
            if cur[a] == r:
                tally[a, : i // checkpoints + 1] = ids[i]
            cur[a] = ids[i]
        if i % checkpoints == 0:
            tally[:, i // checkpoints] = cur
    tally[:, n_ck - 1] = cur
    return dict(tally_ids=tally, tally_checkpoints=checkpoints)


def compute_blocked_ids(raw_ids, code, end_bwt_idx, first_runs, sigma, mode=8):
    """src/move_structure_build.cpp:939-1074: per block and character keep the
    last destination id (relative to first_runs[c+1]) seen before the block; rows
    store the distance from that check point; halve the block size until the
    distance fits MAX_ALLOWED_BLOCKED_ID."""
    r = len(raw_ids)
    fr = np.asarray(first_runs, np.int64)
    block_size, max_allowed = BLOCK_SIZE[mode], MAX_ALLOWED_BLOCKED_ID[mode]
    adj = raw_ids - fr[code + 1]
    not_end = np.arange(r) != end_bwt_idx
    while True:
        nblocks = (r + block_size - 1) // block_size
        id_blocks = np.zeros((sigma, nblocks), np.uint32)
        for a in range(sigma):
            m = (code == a) & not_end
            idx = np.flatnonzero(m)
            # last adjusted id of character a strictly before each block boundary
            k = np.searchsorted(idx, np.arange(nblocks) * block_size, side="left")
            vals = np.where(k > 0, adj[idx[np.maximum(k - 1, 0)]] if len(idx) else 0, 0)
            id_blocks[a] = vals.astype(np.uint32)
        blocked = adj - id_blocks[code, np.arange(r) // block_size].astype(np.int64)
        blocked[~not_end] = 0
        if blocked.max() > max_allowed:
            block_size //= 2
            max_allowed = ((max_allowed + 1) // 2) - 1
            continue
        assert blocked.min() >= 0
        return dict(blocked_id=blocked, id_blocks=id_blocks, block_size=block_size)


def encode_rows(f):
    """Packed rows: include/move_row.hpp:128-142 with the masks of
    include/move_row_configs.hpp:34-51 (mode 6) / :76-104 (mode 8)."""
    r, mode = f["r"], f["mode"]
    n, off, c, t = f["lens"], f["offset"], f["code"], f["thr_bits"]
    if mode == 6:
        rows = np.zeros((r, 4), np.uint16)
        pid = f["pp_id"]
        rows[:, 0] = pid & 0xFFFF
        rows[:, 1] = (pid >> 16) & 0xFFFF
        rows[:, 2] = n | (t[:, 1] << 11) | (t[:, 2] << 12) | (c << 13)
        rows[:, 3] = off | (t[:, 0] << 11) | ((pid >> 32) << 12)
    elif mode == 3:
        # regular, no thresholds (move_row_configs.hpp:21-32): the mode-6 fields with 12-bit n / offset and no threshold bits
        rows = np.zeros((r, 4), np.uint16)
        pid = f["pp_id"]
        rows[:, 0] = pid & 0xFFFF
        rows[:, 1] = (pid >> 16) & 0xFFFF
        rows[:, 2] = n | (c << 13)
        rows[:, 3] = off | ((pid >> 32) << 12)
    elif mode == 2:
        # blocked, no thresholds (move_row_configs.hpp:54-75, MoveRow::set_id src/move_row.cpp:213-225): 24-bit blocked id =
        # id16 | 6 bits in n | 2 bits in offset[15:14]
        rows = np.zeros((r, 3), np.uint16)
        bid = f["blocked_id"]
        rows[:, 0] = bid & 0xFFFF
        rows[:, 1] = n | (((bid >> 16) & 0x3F) << 10)
        rows[:, 2] = off | (c << 10) | ((bid >> 22) << 14)
    elif mode == 5:
        # sampled, no thresholds (move_row_configs.hpp:107-118): u8 n | u8 offset | u8 c with bits 0-1 = offset bits 8-9,
        # bits 2-3 = n bits 8-9, bits 4-7 = character
        rows = np.zeros((r, 3), np.uint8)
        rows[:, 0] = n & 0xFF
        rows[:, 1] = off & 0xFF
        rows[:, 2] = (off >> 8) | ((n >> 8) << 2) | (c << 4)
        return rows.tobytes()
    elif mode == 7:
        # include/move_row.hpp:122-127 + move_row_configs.hpp:120-136: u8 n | u8 offset | u8 c with
        # bit0 = offset bit 8, bit1 = n bit 8, bits 2-4 = character, bits 5-7 = threshold bits 0-2
        rows = np.zeros((r, 3), np.uint8)
        rows[:, 0] = n & 0xFF
        rows[:, 1] = off & 0xFF
        rows[:, 2] = (off >> 8) | ((n >> 8) << 1) | (c << 2) | (t[:, 0] << 5) | (t[:, 1] << 6) | (t[:, 2] << 7)
        return rows.tobytes()
    else:
        rows = np.zeros((r, 3), np.uint16)
        bid = f["blocked_id"]
        rows[:, 0] = bid & 0xFFFF
        rows[:, 1] = n | ((bid >> 16) << 10)
        rows[:, 2] = off | (c << 10) | (t[:, 0] << 13) | (t[:, 1] << 14) | (t[:, 2] << 15)
    return rows.astype("<u2").tobytes()


def serialize(f):
    """src/move_structure_io.cpp:435-469 (v2 header include/utils.hpp:32-61).
    Header padding bytes are uninitialised in the reference; zeros here."""
    u64 = lambda xs: np.asarray(xs, "<u8").tobytes()
    hdr = bytearray(48)
    hdr[0:4] = np.uint32(MOVI_MAGIC).tobytes()
    hdr[4], hdr[5], hdr[6], hdr[7], hdr[8] = 2, 0, 0, f["mode"], 0
    hdr[16:48] = u64([f["n"], f["r"], f["original_r"], f["end_bwt_idx"]])
    out = [bytes(hdr), u64(f["end_thr"]), u64([0] * 4), u64([0] * 4),
           u64([256]), u64(f["alphamap"]), u64([len(f["alphabet"])]), f["alphabet"],
           b"\x00\x00", b"\x00",                       # u16 nt_splitting, bool constant
           encode_rows(f)]
    if f["mode"] in (5, 7):
        # write_tally_table, src/move_structure_io.cpp:328-336: u32 checkpoints | u64 len | per character len x MoveTally
        # (40-bit id: u32 low | u8 high, include/move_row.hpp:13-40)
        tl = f["tally_ids"]
        packed = np.zeros(tl.shape + (5,), np.uint8)
        for b in range(5):
            packed[..., b] = (tl >> (8 * b)) & 0xFF
        out += [np.uint32(f["tally_checkpoints"]).tobytes(), u64([tl.shape[1]]), packed.tobytes()]
    out += [u64([0]), u64([0]), u64([0]),               # overflow tables (empty)
           u64([len(f["counts"])]), u64(f["counts"]),
           u64([len(f["last_runs"])]), u64(f["last_runs"]), u64(f["last_offsets"]),
           u64(f["first_runs"]), u64(f["first_offsets"])]
    if f["mode"] in (8, 2):
        ib = f["id_blocks"]
        out += [u64([ib.shape[1]]), ib.astype("<u4").tobytes(), u64([f["block_size"]])]
    if f.get("sep") and f["mode"] in THRESHOLD_MODES:
        # write_separators_thresholds, src/move_structure_io.cpp:399-413: u64 count | ThresholdsRow{u16[4]} each |
        # u64 map size | (u64 row, u64 entry) pairs.  The reference walks an unordered_map (unspecified order);
        # ascending row order here.
        st = np.asarray(f["sep_thr"], "<u2").reshape(-1, 4)
        out += [u64([len(st)]), st.tobytes(), u64([len(f["sep_map"])])]
        out += [u64([k, v]) for k, v in sorted(f["sep_map"].items())]
    return b"".join(out)


def build_index_from_seqs(seqs, mode, rc=True, separators=False):
    t = clean_text(seqs, rc=rc, separators=separators)
    bwt, thr = bwt_and_thresholds(t)
    return serialize(build_rows(bwt, thr, mode))


def build_index_from_fasta(path, mode, separators=False):
    return build_index_from_seqs([s for _, s in read_fasta(path)], mode, separators=separators)


if __name__ == "__main__":
    import sys, os
    fasta, mode, outdir = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    os.makedirs(outdir, exist_ok=True)
    data = build_index_from_fasta(fasta, mode)
    with open(os.path.join(outdir, "index.movi"), "wb") as fo:
        fo.write(data)
    print(len(data))
