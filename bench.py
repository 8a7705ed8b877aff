#!/usr/bin/env python3
"""bench.py -- PML query throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the PML hot path (HIP kernel behind the C-ABI) over one
batch of synthetic reads already resident in HBM.  Default workload = BASELINE
config[1]: a synthetic ~10 M-row regular-thresholds (mode 6) table and
1 M x 150 bp reads per GPU.  With N > 1 (launched by torch.distributed.run, one
rank per GPU) rank 0 builds the index, its row table is broadcast once over
RCCL/xGMI, every rank queries its own shard of reads, no data-path collective
("scaling": "weak").

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

PG_C2 = dict(anc=5_000_000, genomes=64, snp=0.001, seed=11)    # 64 x 5 Mbp (+ reverse complements) = 640 Mbp, ~14 M rows
PG = PG_C2
WORKLOADS = {
    # "pangenome": real BWT of a synthetic 64-genome pangenome built by tools/build_index,
    # cached under /tmp after the first ~90 s build.  "synth": random move table of tools/synth.c (8(d)(ii)).
    "c2": dict(kind="pangenome", mode=6, reads=1_000_000, read_len=150, sub=0.01,
               desc="BASELINE config 2: synthetic 64-genome E. coli-scale pangenome (5 Mbp ancestor, 640 Mbp text, ~14 M "
                    "rows, real BWT), regular-thresholds, 1M x 150bp reads per GPU (substrings, 1% subst., 0.1% N)"),
    "c3": dict(kind="pangenome", mode=6, reads=100_000, read_len=10_000, sub=0.08,
               desc="BASELINE config 3: same pangenome index, 100k x 10kbp reads per GPU (8% subst.)"),
    "c2b": dict(kind="pangenome", mode=8, reads=1_000_000, read_len=150, sub=0.01,
                desc="same pangenome as blocked-thresholds (6 B rows), 1M x 150bp reads per GPU"),
    "c2s": dict(kind="pangenome", mode=7, reads=1_000_000, read_len=150, sub=0.01,
                desc="same pangenome as sampled-thresholds (3 B rows + id checkpoints every 20 rows on disk; expanded to "
                     "regular-thresholds rows on the GPU at upload), 1M x 150bp reads per GPU"),
    "c2synth": dict(kind="synth", rows=10_000_000, mode=6, reads=1_000_000, read_len=150, sub=0.01,
                    desc="random 10M-row regular-thresholds table (worst-case step mix), 1M x 150bp reads per GPU"),
    "c3synth": dict(kind="synth", rows=10_000_000, mode=6, reads=100_000, read_len=10_000, sub=0.08,
                    desc="random 10M-row regular-thresholds table, 100k x 10kbp reads per GPU"),
    "c2bsynth": dict(kind="synth", rows=10_000_000, mode=8, reads=1_000_000, read_len=150, sub=0.01,
                     desc="random 10M-row blocked-thresholds table, 1M x 150bp reads per GPU"),
    "c4": dict(kind="synth", rows=1_000_000_000, mode=6, reads=1_250_000, read_len=150, sub=0.01,
               desc="random 1B-row regular-thresholds table (8 GB), 1.25M x 150bp reads per GPU (BASELINE config 4 shard)"),
    "c5": dict(kind="synth", rows=1_000_000_000, mode=8, reads=1_250_000, read_len=150, sub=0.01,
               desc="random 1B-row blocked-thresholds table (6 GB), 1.25M x 150bp reads per GPU (BASELINE config 5 shard; "
                    "use with --query count)"),
    "c2mid": dict(kind="pangenome", pg="mid", mode=6, reads=1_000_000, read_len=150, sub=0.01,
                  desc="a real BWT that fits NO cache and still builds in a minute or two (round 6): 64 genomes of a 2.5 Mbp ancestor, 1.3 % SNPs "
                       "(320 Mbp text, 39.5 M rows = 316 MB; look-ahead copy 632 MB, deep rows 842 MB: beyond the 256 MB Infinity Cache), 1M x 150bp reads per GPU"),
    "c4real": dict(kind="pangenome", pg="big", mode=6, reads=1_250_000, read_len=150, sub=0.01,
                   desc="real BWT beyond the Infinity Cache and the TLBs' reach: synthetic 64-genome pangenome, 8.5 Mbp ancestor, 1 % "
                        "SNPs (1.09 Gbp text, ~102 M rows = 0.8 GB, built on first use: ~10 min), 1.25M x 150bp reads per GPU"),
    "c4real2": dict(kind="pangenome", pg="big2", mode=6, reads=1_250_000, read_len=150, sub=0.01,
                    desc="the same with a 16.5 Mbp ancestor: 2.11 Gbp text, ~220 M rows = 1.8 GB (look-ahead copy 3.5 GB), ~6 min to build"),
    "c4big": dict(kind="pangenome", pg="big3", mode=6, reads=1_250_000, read_len=150, sub=0.01,
                  desc="a real BWT of two thirds of a billion rows (round 5): the 2.11 Gbp pangenome with 4 % SNPs between its 64 genomes (n / r = 3.1: "
                       "676 M rows = 5.4 GB, look-ahead copy 10.8 GB), ~7 min to build, 1.25M x 150bp reads per GPU"),
    "tiny": dict(kind="synth", rows=200_000, mode=6, reads=20_000, read_len=150, sub=0.01,
                 desc="tiny plumbing workload"),
    "tinypg": dict(kind="pangenome", pg="tiny", mode=6, reads=20_000, read_len=150, sub=0.01,
                   desc="tiny plumbing workload on a real BWT (8 x 60 kbp pangenome): runs the default line's legs in tests"),
}
PG_BIG = dict(anc=8_500_000, genomes=64, snp=0.01, seed=12)   # ~102 M rows (n / r = 10.7): tools/build_index, ~10 min, ~16 GB of host memory
PG_BIG2 = dict(anc=16_500_000, genomes=64, snp=0.01, seed=14)  # ~220 M rows: the largest text the 32-bit suffix array takes (2.11 Gbp), ~35 GB of host memory
PG_BIG3 = dict(anc=16_500_000, genomes=64, snp=0.04, seed=15)  # the same 2.11 Gbp with 4 % SNPs: n / r = 3.1 -> 675 738 185 rows (0.68 B): a REAL BWT at the size of the BASELINE target's table (small-scale calibration of n / r against the SNP rate: 1 % 11.1, 3 % 4.6, 5 % 3.1, 8 % 2.3)
PG_TINY = dict(anc=60_000, genomes=8, snp=0.002, seed=13)     # tests: built in a second
PG_MID = dict(anc=2_500_000, genomes=64, snp=0.013, seed=16)  # 39 493 027 rows (n / r = 8.1; 320 Mbp of text): tools/build_index in 1 - 3 min, ~5 GB of host memory


def pg_of(wl):
    return {"big": PG_BIG, "big2": PG_BIG2, "big3": PG_BIG3, "tiny": PG_TINY, "mid": PG_MID}.get(wl.get("pg"), PG_C2)
# where built pangenome indexes and their reads are kept: $MOVI_BENCH_CACHE, else a .bench_cache/ beside this file if one
# travelled with the tree (a prebuilt c2 index saves the ~2 min single-threaded build per fresh box), else /tmp
CACHE = os.environ.get("MOVI_BENCH_CACHE") or (os.path.join(ROOT, ".bench_cache") if os.path.isdir(os.path.join(ROOT, ".bench_cache"))
                                                 else "/tmp/movi_bench_cache")


def ensure_pangenome(wl, world, rank, barrier):
    """Rank 0 builds (once per box) the pangenome index and this workload's reads; returns the directory."""
    import subprocess
    tool = os.path.join(ROOT, "tools", "build_index")
    PG = pg_of(wl)
    idx_dir = os.path.join(CACHE, "pg_%d_%d_%g_%d_m%d" % (PG["anc"], PG["genomes"], PG["snp"], PG["seed"], wl["mode"]))
    reads_file = os.path.join(idx_dir, "reads_%dx%d_%g.bin" % (wl["reads"] * world, wl["read_len"], wl["sub"]))
    if rank == 0:
        src = tool + ".cpp"
        if not os.path.exists(tool) or os.path.getmtime(tool) < os.path.getmtime(src) or not os.access(tool, os.X_OK):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", tool, src])
        os.makedirs(CACHE, exist_ok=True)
        if not os.path.exists(os.path.join(idx_dir, ".done")):
            tmp = idx_dir + ".tmp%d" % os.getpid()
            subprocess.check_call([tool, "pangenome", str(PG["anc"]), str(PG["genomes"]), str(PG["snp"]), str(PG["seed"]),
                                   str(wl["mode"]), tmp], stderr=subprocess.DEVNULL)
            if os.path.exists(idx_dir):
                import shutil
                shutil.rmtree(idx_dir)
            os.rename(tmp, idx_dir)
            open(os.path.join(idx_dir, ".done"), "w").close()
        if not os.path.exists(reads_file):
            if not os.path.exists(os.path.join(idx_dir, "text.bin")):      # a cache that travelled without its text (seconds to redo)
                subprocess.check_call([tool, "pangenome", str(PG["anc"]), str(PG["genomes"]), str(PG["snp"]), str(PG["seed"]),
                                       str(wl["mode"]), idx_dir, "text-only"], stderr=subprocess.DEVNULL)
            subprocess.check_call([tool, "reads", os.path.join(idx_dir, "text.bin"), str(wl["reads"] * world),
                                   str(wl["read_len"]), str(wl["sub"]), str(PG["seed"]), reads_file + ".tmp"])
            os.rename(reads_file + ".tmp", reads_file)
    barrier()
    return idx_dir, reads_file


def served_from(table_bytes):
    """Where the walk's row gathers are served from in steady state: a table (as walked) that fits the 256 MiB Infinity Cache
    hardly reaches DRAM -- the line rate of the L2 <-> fabric path bounds it -- anything bigger comes from HBM.  Reported
    beside the roofline; `bound` itself stays "hbm" (the contract's label: an HBM-bandwidth roofline, `peak` = the HBM peak)."""
    return "hbm" if table_bytes > INFINITY_CACHE_BYTES else "infinity cache (fabric line rate)"


def lookup_traffic(key, rows, reads, read_len, kernel):
    """roofline.traffic: bytes per launch from profiles/traffic.json -- rocprofv3 PMC passes of an EARLIER run of this very
    launch (counters cannot be collected inside a plain bench run), not a measurement of this run; null when the shape or
    the kernel differs from what was profiled.  `key` names the launch ("c2", "c3", "c3_classify1", "c4", "c4_count", ...);
    `kernel` is the library's name for it with EVERY template argument (movi_last_launch), and it must equal the entry's."""
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        ent = json.load(open(tf)).get(key)
        if ent and ent.get("rows") == rows and ent.get("reads") == reads and ent.get("read_len") == read_len \
                and ent.get("kernel_name") == kernel:
            return ent.get("hbm_bytes_per_launch"), "static: %s (PMC passes of an earlier run of this launch; not measured in this run)" % ent.get("source")
    except Exception:
        pass
    return None, None


def timed_steps(torch, dist, world, dev, stream, run, steps):
    """Exactly `steps` calls of run() bracketed by barrier + synchronize on both sides.  Returns (seconds between the
    barriers: the max over ranks, [every rank's own seconds up to its synchronize], this rank's mean kernel seconds from HIP
    events recorded on the launch stream)."""
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record(stream)
        run()
        b.record(stream)
    torch.cuda.synchronize()
    own = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    per = [own]
    if world > 1:
        t = torch.tensor([el, own], dtype=torch.float64, device=dev)
        lst = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(lst, t)
        el = max(float(x[0]) for x in lst)
        per = [float(x[1]) for x in lst]
    return el, per, sum(a.elapsed_time(b) for a, b in evs) / steps / 1e3


def sum_over_ranks(torch, dist, world, dev, x):
    if world == 1:
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


SIDE_TABLE_BYTES = 16 << 24       # the 4^12 x 16 B top-of-walk / interval table: one lookup per read (~0.1 B per base), reported beside the roofline


def pml_roofline(table_bytes, row_bytes, st, n_bases, kern_s, launch, traffic=None, tsrc=None):
    """The `roofline` object of a PML leg: algorithmic bytes (row_bytes x (1 + f + s) + 1 + 2 per base, with the
    REFERENCE's row size -- the look-ahead copy's 16 B per row are this engine's layout, not the algorithm's) over the kernel's
    mean launch time, against the HBM peak.  `bound` follows the bytes the row gathers walk (`table_bytes`: the look-ahead copy
    where the launch walked on it); the side table is reported separately."""
    f_bar, s_bar = st.fast_forwards / max(n_bases, 1), st.scans / max(n_bases, 1)
    bpb = row_bytes * (1.0 + f_bar + s_bar) + 1 + 2
    ach = bpb * n_bases / kern_s / 1e9
    return {"bound": "hbm", "gathers_served_from": served_from(table_bytes), "working_set_bytes": table_bytes, "side_table_bytes": SIDE_TABLE_BYTES,
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": tsrc, "algorithmic_bytes_per_base": round(bpb, 3),
            "algorithmic_model": "SURVEY 8(d): %d B (the reference's row) x rows the reference's walk reads per base + 3" % row_bytes,
            "kernel": launch["kernel"], "launch": launch, "kernel_ms_avg": kern_s * 1e3,
            "lane_iterations_per_s": st.lane_steps / kern_s if st.wave_steps else None,
            "rows_read_per_s": (1.0 + f_bar + s_bar) * n_bases / kern_s}


def count_roofline(table_bytes, row_bytes, st, matched_bases, kern_s, launch, traffic=None, tsrc=None):
    """The same for the count query: B_count = 2 x row_bytes x (1 + f) + row_bytes x u + 1 per base the search extends over
    (two LF walkers, u interval-shrink rows, one base in)."""
    wb = max(int(matched_bases), 1)
    f_bar, u_bar = st.fast_forwards / wb / 2.0, st.scans / wb
    bpb = 2 * row_bytes * (1.0 + f_bar) + row_bytes * u_bar + 1
    ach = bpb * wb / kern_s / 1e9
    return {"bound": "hbm", "gathers_served_from": served_from(table_bytes), "working_set_bytes": table_bytes, "side_table_bytes": SIDE_TABLE_BYTES,
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": tsrc, "algorithmic_bytes_per_base": round(bpb, 3), "matched_bases_per_step": wb,
            "kernel": launch["kernel"], "launch": launch, "kernel_ms_avg": kern_s * 1e3,
            "lane_iterations_per_s": st.lane_steps / kern_s if st.wave_steps else None}


def table_bytes_walked(rows, row_bytes, launch):
    """Bytes of the table the launch gathered from: the look-ahead copy is 16 B per row, the deep rows 64 B per three rows."""
    return int(rows * {0: row_bytes, 1: 16, 2: 64.0 / 3.0}.get(int(launch.get("ahead") or 0), row_bytes))


def send_to_rank(torch, dist, world, rank, dev, arr, dst):
    """Rank 0 hands numpy array `arr` to rank `dst` (one broadcast per hand-over: works on every backend); returns it there."""
    meta = [(arr.shape, str(arr.dtype)) if rank == 0 else None]
    dist.broadcast_object_list(meta, src=0)
    shape, dt = meta[0]
    t = torch.from_numpy(arr.view(np.uint8).reshape(-1)).to(dev) if rank == 0 else \
        torch.empty(int(np.prod(shape)) * np.dtype(dt).itemsize, dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=0)
    return t.cpu().numpy().view(dt).reshape(shape) if rank == dst else None


def draw_synth_reads(torch, dist, world, rank, dev, synth, six, n_reads, read_len, seed, sub, lens_of=None):
    """Reads for a tools/synth.c table: rank 0 (the only rank that holds the table's host structure) draws every rank's
    shard -- seed + rank -- and hands it over; no other rank synthesises anything."""
    mine = None
    for p in range(world):
        if rank == 0:
            lens = lens_of(p) if lens_of else None
            b, o = synth.synth_reads(six, n_reads, read_len, seed=seed + p, sub_rate=sub, n_rate=0.001, lens=lens)
        if world == 1:
            return b, o
        bb = send_to_rank(torch, dist, world, rank, dev, b if rank == 0 else None, p)
        oo = send_to_rank(torch, dist, world, rank, dev, o if rank == 0 else None, p)
        if rank == p:
            mine = (bb, oo)
    return mine


def big_table_leg(torch, dist, world, rank, dev, local_rank, stream, synth, movi_amd, cores, rows=1_000_000_000, steps=10, warmup=2, host_legs=True):
    """c4: tools/synth.c's 1 B-row regular-thresholds table (8 GB), 1.25 M x 150 bp reads per GPU.  Rank 0 synthesises the
    table and draws every rank's reads; the rows reach the other GPUs through ONE broadcast (RCCL).  Returns the `big_table`
    object of the bench line on rank 0 (None elsewhere).  The oracle is the checker here, never the thing measured."""
    from oracle.oracle import Oracle
    from movi_amd._lib import IndexDescC
    w = WORKLOADS["c4"]
    n_reads, L = w["reads"], w["read_len"]
    t0 = time.time()
    six = img = meta = d_rows = None
    if rank == 0:
        six = synth.synth_index(rows, mode=6, seed=SEED)
        img = six.image()
        _, cdesc0, roff, rbytes = movi_amd.parse_index_image(img)
        meta = {"cdesc": bytes(cdesc0)}
    t_gen = time.time() - t0
    t0 = time.time()
    if rank == 0:
        d_rows = torch.from_numpy(img[roff: roff + rbytes]).to(dev)
    torch.cuda.synchronize()
    t_up = time.time() - t0
    t_bc = 0.0
    if world > 1:
        from movi_amd import dist as md
        tb = time.time()
        meta, d_rows = md.broadcast_index(meta, d_rows, src=0, device=dev)       # the one collective of this path
        torch.cuda.synchronize()
        t_bc = time.time() - tb
    cdesc = IndexDescC.from_buffer_copy(meta["cdesc"])
    cdesc.id_blocks = None
    cdesc.tally_ids = None
    cdesc.separator_thresholds, cdesc.separator_map = None, None
    rbytes = int(d_rows.numel())
    index = movi_amd.MoveIndex.from_device_rows(cdesc, d_rows.data_ptr(), device=local_rank, keepalive=d_rows)
    t0 = time.time()
    derived = index.prepare(index.PREPARE_PML | index.PREPARE_COUNT)       # look-ahead copy (16 GB), top-of-walk / interval tables, checkpoints
    t_prep = time.time() - t0
    bases, offs = draw_synth_reads(torch, dist, world, rank, dev, synth, six, n_reads, L, SEED + 1, w["sub"])
    n_bases = int(bases.size)
    d_bases = torch.from_numpy(bases).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
    d_out = torch.empty(n_bases, dtype=torch.int16, device=dev)
    d_err = torch.zeros(n_reads, dtype=torch.uint8, device=dev)
    run = lambda: index.pml_device(d_bases.data_ptr(), d_offs.data_ptr(), n_reads, n_bases, d_out.data_ptr(), d_err.data_ptr(),
                                   stream.cuda_stream, 0)
    for _ in range(warmup):
        run()
    torch.cuda.synchronize()
    st = index.last_stats(stream.cuda_stream)
    dt, per, kern_s = timed_steps(torch, dist, world, dev, stream, run, steps)
    total = sum_over_ranks(torch, dist, world, dev, n_bases)
    launch = index.last_launch()
    traffic, tsrc = lookup_traffic("c4", rows, n_reads, L, launch["kernel"])
    roof = pml_roofline(table_bytes_walked(rows, 8, launch), 8, st, n_bases, kern_s, launch, traffic, tsrc)
    roof["dram_frac_of_peak"] = (traffic / kern_s / 1e9 / HBM_PEAK_GBS) if traffic else None
    out = {"workload": "c4", "description": w["desc"], "value": total * steps / dt / 1e9, "unit": "Gbases/s", "n_gpus": world, "steps": steps,
           "warmup": warmup, "ms_per_step": dt / steps * 1e3, "rank_seconds": [round(x, 4) for x in per],
           "rows": rows, "table_bytes": int(rbytes), "reads_per_gpu": n_reads,
           "read_len": L, "fast_forwards_per_base": round(st.fast_forwards / n_bases, 4), "scans_per_base": round(st.scans / n_bases, 4),
           "simt_efficiency": round(st.lane_steps / (64.0 * st.wave_steps), 4) if st.wave_steps else None,
           "iterations_per_base": round(st.lane_steps / n_bases, 4) if st.wave_steps else None,
           "algorithmic_bytes_per_base": roof["algorithmic_bytes_per_base"], "errors": int(st.errors),
           "index_gen_s": round(t_gen, 1), "index_upload_s": round(t_up, 2), "index_broadcast_s": round(t_bc, 3),
           "prepare_s": round(t_prep, 3), "derived_bytes": derived, "roofline": roof}
    # BASELINE config 5's query on the same resident rows (a blocked-thresholds file of this table expands to exactly them at
    # upload: tests/test_big_table_gpu.py runs that form): --count
    d_m = torch.zeros(n_reads, dtype=torch.int64, device=dev)
    d_c = torch.zeros(n_reads, dtype=torch.int64, device=dev)
    runc = lambda: index.count_device(d_bases.data_ptr(), d_offs.data_ptr(), n_reads, n_bases, d_m.data_ptr(), d_c.data_ptr(),
                                      d_err.data_ptr(), stream.cuda_stream, 0)
    runc()
    torch.cuda.synchronize()
    stc = index.last_stats(stream.cuda_stream)
    dtc, perc, kern_c = timed_steps(torch, dist, world, dev, stream, runc, 5)
    claunch = index.last_launch()
    ctraffic, ctsrc = lookup_traffic("c4_count", rows, n_reads, L, claunch["kernel"])
    out["count"] = {"config": {"mode": 6, "note": "--count on the regular-thresholds (8-byte) rows of this leg's table; BASELINE config 5 as worded -- the "
                                                   "blocked-thresholds file, 6-byte rows expanded on the GPU -- is `count_blocked_thresholds` below"},
                    "value": total * 5 / dtc / 1e9, "unit": "Gbases/s (read bases)", "steps": 5, "ms_per_step": dtc / 5 * 1e3,
                    "rank_seconds": [round(x, 4) for x in perc], "kernel": claunch["kernel"],
                    "roofline": count_roofline(table_bytes_walked(rows, 8, claunch), 8, stc, int(d_m.sum().item()), kern_c, claunch, ctraffic, ctsrc)}
    out["count"]["roofline"]["dram_frac_of_peak"] = (ctraffic / kern_c / 1e9 / HBM_PEAK_GBS) if ctraffic else None
    out["zml"] = zml_leg(torch, dist, world, dev, stream, index, rows, 8, d_bases, d_offs, n_reads, n_bases, d_out, d_err, "c4_zml", L)
    if getattr(index, "dry", False):
        out["dry_run"] = True
        out["index_broadcast_gb_s"] = round(rbytes / t_bc / 1e9, 2) if t_bc > 0 else None
        index.close()
        c8 = count_mode8_leg(torch, dist, world, rank, dev, local_rank, stream, synth, movi_amd, cores, rows)
        if rank == 0:
            out["count_blocked_thresholds"] = c8
        return out if rank == 0 else None
    out["index_broadcast_gb_s"] = round(rbytes / t_bc / 1e9, 2) if t_bc > 0 else None
    if rank == 0:
        # parity: three slices of rank 0's batch against the oracle on the same image, PMLs, counters and counts
        t0 = time.time()
        cpu = Oracle(img)
        run()
        torch.cuda.synchronize()
        got_all = d_out.cpu().numpy().view(np.uint16)
        gm, gc = d_m.cpu().numpy().view(np.uint64), d_c.cpu().numpy().view(np.uint64)
        ok, cok = True, True
        for lo in (0, n_reads // 2 - 1000, n_reads - 2000):
            hi = lo + 2000
            sb = bases[int(offs[lo]): int(offs[hi])]
            so = offs[lo: hi + 1] - offs[lo]
            exp, eff, esc = cpu.pml_batch(sb, so, threads=cores)
            ok = ok and bool((got_all[int(offs[lo]): int(offs[hi])] == exp).all())
            # the same slice alone through the engine: its fast-forward / scan counters must equal the oracle's
            sl = torch.from_numpy(np.ascontiguousarray(so).view(np.int64)).to(dev)
            index.pml_device(d_bases.data_ptr() + int(offs[lo]), sl.data_ptr(), 2000, int(sb.size), d_out.data_ptr(), d_err.data_ptr(),
                             stream.cuda_stream, 0)
            sst = index.last_stats(stream.cuda_stream)
            ok = ok and sst.fast_forwards == eff and sst.scans == esc and sst.errors == 0
            ok = ok and bool((d_out[: sb.size].cpu().numpy().view(np.uint16) == exp).all())
            em, ec = cpu.count_batch(sb, so, threads=cores)
            cok = cok and bool((gm[lo:hi] == em).all() and (gc[lo:hi] == ec).all())
        out["count"]["matched_bases_per_read"] = round(float(gm.mean()), 2)
        out["count"]["parity_sample_ok"] = cok
        out["parity_sample_ok"] = ok and cok
        out["parity_sample"] = "rank 0's reads [0,2000), [%d,%d), [%d,%d) vs oracle/movi_oracle.c on the same image: PMLs bit-exact, fast-forward and scan counters equal; count: matched lengths and counts equal (%.1f s)" % (
            n_reads // 2 - 1000, n_reads // 2 + 1000, n_reads - 2000, n_reads, time.time() - t0)
        cpu.close()
    if rank == 0 and world == 1 and host_legs:
        # round 5: the PCIe-inclusive and the command-line rates on THIS table too (the table the BASELINE target is quoted on)
        try:
            from movi_amd._lib import QueryStatsC, check, lib
            hp = {"unit": "Gbases/s", "note": "movi_pml_host on the 1 B-row table, best of 3 calls after a warm-up call, checked against the device call's vector"}
            stq = QueryStatsC()
            ref = got_all                                         # the device call's vector of the whole batch (fetched before the slices re-used d_out)
            for name, mk in (("pageable", lambda n, dt: np.empty(n, dt)), ("page_locked", movi_amd.pinned_empty)):
                hb, ho = mk(n_bases, np.uint8), mk(n_bases, np.uint16)
                hb[:] = bases
                ts = []
                for _ in range(4):
                    t1 = time.perf_counter()
                    check(lib().movi_pml_host(index._h, hb.ctypes.data, np.ascontiguousarray(offs, np.uint64).ctypes.data, n_reads, ho.ctypes.data, None, C.byref(stq)))
                    ts.append(time.perf_counter() - t1)
                hp[name] = round(n_bases / min(ts[1:]) / 1e9, 2)
                hp[name + "_ok"] = bool((ho == ref).all())
                del hb, ho
            out["host_path"] = hp
        except Exception as e:                            # noqa: BLE001
            out["host_path"] = {"error": repr(e)[:200]}
    index.close()
    if rank == 0 and world == 1 and host_legs:
        import shutil
        import tempfile
        tmp = tempfile.mkdtemp(prefix="movi_cli_big_")
        try:
            img.tofile(os.path.join(tmp, "index.movi"))           # 8 GB: the command maps it like any index file
            fa = os.path.join(tmp, "reads.fa")
            write_fasta(fa, bases.reshape(-1, L))
            out["cli_path"] = dict(cli_measure(tmp, fa, os.path.join(tmp, "out")), unit="Gbases/s",
                                   note="movi query on the 1 B-row index file (8 GB, mapped) and a FASTA file of this leg's reads; value = bases / the command's own read-processing time")
        except Exception as e:                            # noqa: BLE001
            out["cli_path"] = {"error": repr(e)[:300]}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    # ---- config 5 as worded: the blocked-thresholds form of the table (6 GB of file rows up, expanded on the GPU), --count
    del six, img
    torch.cuda.empty_cache()
    try:
        c8 = count_mode8_leg(torch, dist, world, rank, dev, local_rank, stream, synth, movi_amd, cores, rows)
        if rank == 0:
            out["count_blocked_thresholds"] = c8
            if c8.get("parity_sample_ok") is False:
                out["parity_sample_ok"] = False
    except Exception as e:                                # noqa: BLE001
        if world > 1:
            raise
        out["count_blocked_thresholds"] = {"error": repr(e)[:300]}
    return out if rank == 0 else None


def zml_roofline(table_bytes, row_bytes, st, n_bases, kern_s, launch, traffic=None, tsrc=None):
    """The `roofline` object of a ZML leg: every base costs the two LF walkers of the interval's ends (fast-forwards counted over both), the
    interval-shrink rows, one base in and one u16 out: B_zml = 2 x row_bytes x (1 + f) + row_bytes x u + 1 + 2 (as B_count + the output)."""
    f_bar, u_bar = st.fast_forwards / max(n_bases, 1) / 2.0, st.scans / max(n_bases, 1)
    bpb = 2 * row_bytes * (1.0 + f_bar) + row_bytes * u_bar + 1 + 2
    ach = bpb * n_bases / kern_s / 1e9
    return {"bound": "hbm", "gathers_served_from": served_from(table_bytes), "working_set_bytes": table_bytes, "side_table_bytes": 0,
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": tsrc, "algorithmic_bytes_per_base": round(bpb, 3),
            "kernel": launch["kernel"], "launch": launch, "kernel_ms_avg": kern_s * 1e3,
            "lane_iterations_per_s": st.lane_steps / kern_s if st.wave_steps else None}


def zml_leg(torch, dist, world, dev, stream, index, rows, row_bytes, d_bases, d_offs, n_reads, n_bases, d_out, d_err, key, read_len, steps=5):
    """`--zml` (MoveStructure::query_zml, src/move_structure_query.cpp:690-785) on a batch that is already resident, with its roofline
    object (round 6: the ZML parse measured in the default line like the PML walk and the count query)."""
    run = lambda: index.zml_device(d_bases.data_ptr(), d_offs.data_ptr(), n_reads, n_bases, d_out.data_ptr(), d_err.data_ptr(), stream.cuda_stream, 0)
    run()
    torch.cuda.synchronize()
    st = index.last_stats(stream.cuda_stream)
    dt, per, kern_s = timed_steps(torch, dist, world, dev, stream, run, steps)
    total = sum_over_ranks(torch, dist, world, dev, n_bases)
    launch = index.last_launch()
    traffic, tsrc = lookup_traffic(key, rows, n_reads, read_len, launch["kernel"])
    roof = zml_roofline(table_bytes_walked(rows, row_bytes, launch), row_bytes, st, n_bases, kern_s, launch, traffic, tsrc)
    roof["dram_frac_of_peak"] = (traffic / kern_s / 1e9 / HBM_PEAK_GBS) if traffic else None
    return {"value": total * steps / dt / 1e9, "unit": "Gbases/s", "steps": steps, "ms_per_step": dt / steps * 1e3, "rank_seconds": [round(x, 4) for x in per],
            "kernel": launch["kernel"], "errors": int(st.errors),
            "simt_efficiency": round(st.lane_steps / (64.0 * st.wave_steps), 4) if st.wave_steps else None,
            "iterations_per_base": round(st.lane_steps / max(n_bases, 1), 4) if st.wave_steps else None, "roofline": roof}


def count_mode8_leg(torch, dist, world, rank, dev, local_rank, stream, synth, movi_amd, cores, rows, steps=5):
    """BASELINE config 5 AS WORDED (round 6): "the same 1 B-run reference built as blocked-thresholds (6 B/row) ... --count".  The run
    structure of the c4 table as a blocked-thresholds FILE -- 6-byte rows + id blocks (include/move_row.hpp:128-142, src/move_structure.cpp:91-102)
    -- goes up as it is stored (6 GB), is expanded on the GPU to the resident 8-byte layout (`expand_s`), and the count query runs on that.
    `roofline` prices the reference's 6-byte row (B_count); the 8-byte figure of the resident layout is beside it."""
    from oracle.oracle import Oracle
    from movi_amd._lib import IndexDescC
    w = WORKLOADS["c5"]
    n_reads, L = w["reads"], w["read_len"]
    t0 = time.time()
    six = img = meta = d_rows = None
    if rank == 0:
        six = synth.synth_index(rows, mode=8, seed=SEED)
        img = six.image()
        _, cdesc0, roff, rbytes = movi_amd.parse_index_image(img)
        id_blocks = np.ctypeslib.as_array(C.cast(cdesc0.id_blocks, C.POINTER(C.c_uint32)), shape=(int(cdesc0.n_blocks) * int(cdesc0.alphabet_size),)).copy()
        meta = {"cdesc": bytes(cdesc0), "id_blocks": id_blocks}
    t_gen = time.time() - t0
    t0 = time.time()
    if rank == 0:
        d_rows = torch.from_numpy(img[roff: roff + rbytes]).to(dev)
    torch.cuda.synchronize()
    t_up = time.time() - t0
    t_bc = 0.0
    if world > 1:
        from movi_amd import dist as md
        tb = time.time()
        meta, d_rows = md.broadcast_index(meta, d_rows, src=0, device=dev)
        torch.cuda.synchronize()
        t_bc = time.time() - tb
    cdesc = IndexDescC.from_buffer_copy(meta["cdesc"])
    id_blocks = meta["id_blocks"]
    cdesc.id_blocks = id_blocks.ctypes.data
    cdesc.tally_ids = None
    cdesc.separator_thresholds, cdesc.separator_map = None, None
    file_bytes = int(d_rows.numel())
    t0 = time.time()
    index = movi_amd.MoveIndex.from_device_rows(cdesc, d_rows.data_ptr(), device=local_rank, keepalive=d_rows)   # get_id per row, once: 6 B -> 8 B rows
    torch.cuda.synchronize()
    t_expand = time.time() - t0
    del d_rows                                              # (modes 7 / 8: the file rows may be released after the call)
    index._keep = None
    torch.cuda.empty_cache()
    t0 = time.time()
    derived = index.prepare(index.PREPARE_COUNT)
    t_prep = time.time() - t0
    bases, offs = draw_synth_reads(torch, dist, world, rank, dev, synth, six, n_reads, L, SEED + 1, w["sub"])
    n_bases = int(bases.size)
    d_bases = torch.from_numpy(bases).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
    d_err = torch.zeros(n_reads, dtype=torch.uint8, device=dev)
    d_m = torch.zeros(n_reads, dtype=torch.int64, device=dev)
    d_c = torch.zeros(n_reads, dtype=torch.int64, device=dev)
    runc = lambda: index.count_device(d_bases.data_ptr(), d_offs.data_ptr(), n_reads, n_bases, d_m.data_ptr(), d_c.data_ptr(),
                                      d_err.data_ptr(), stream.cuda_stream, 0)
    runc()
    torch.cuda.synchronize()
    stc = index.last_stats(stream.cuda_stream)
    dtc, perc, kern_c = timed_steps(torch, dist, world, dev, stream, runc, steps)
    total = sum_over_ranks(torch, dist, world, dev, n_bases)
    claunch = index.last_launch()
    matched = int(d_m.sum().item())
    ctraffic, ctsrc = lookup_traffic("c4_count", rows, n_reads, L, claunch["kernel"])
    roof6 = count_roofline(rows * 8, 6, stc, matched, kern_c, claunch, ctraffic, ctsrc)
    roof8 = count_roofline(rows * 8, 8, stc, matched, kern_c, claunch, ctraffic, ctsrc)
    roof6["algorithmic_model"] = "SURVEY 8(d) B_count with the reference's 6-byte blocked-thresholds row; the walk itself runs on the 8-byte rows the upload expanded them to"
    roof6["resident_layout"] = {"row_bytes": 8, "algorithmic_bytes_per_base": roof8["algorithmic_bytes_per_base"], "achieved": roof8["achieved"], "frac": roof8["frac"]}
    roof6["dram_frac_of_peak"] = (ctraffic / kern_c / 1e9 / HBM_PEAK_GBS) if ctraffic else None
    out = {"workload": "c5", "description": w["desc"], "config": {"mode": 8, "rows": rows, "file_row_bytes": 6, "file_table_bytes": file_bytes, "resident_row_bytes": 8,
                                                                   "id_blocks": int(id_blocks.size), "reads_per_gpu": n_reads, "read_len": L},
           "value": total * steps / dtc / 1e9, "unit": "Gbases/s (read bases)", "n_gpus": world, "steps": steps, "ms_per_step": dtc / steps * 1e3,
           "rank_seconds": [round(x, 4) for x in perc], "kernel": claunch["kernel"], "matched_bases_per_read": round(matched / n_reads, 2),
           "index_gen_s": round(t_gen, 1), "index_upload_s": round(t_up, 2), "index_broadcast_s": round(t_bc, 3), "expand_s": round(t_expand, 3),
           "expand_gb_s": round(file_bytes / max(t_expand, 1e-9) / 1e9, 1), "prepare_s": round(t_prep, 3), "derived_bytes": derived, "errors": int(stc.errors),
           "roofline": roof6}
    if getattr(index, "dry", False):
        out["dry_run"] = True
    elif rank == 0:
        cpu = Oracle(img)                                    # (the oracle walks the blocked ids as stored: get_id per step)
        gm, gc = d_m.cpu().numpy().view(np.uint64), d_c.cpu().numpy().view(np.uint64)
        ok = True
        for lo in (0, n_reads // 2 - 1000, n_reads - 2000):
            hi = lo + 2000
            sb = bases[int(offs[lo]): int(offs[hi])]
            so = offs[lo: hi + 1] - offs[lo]
            em, ec = cpu.count_batch(sb, so, threads=cores)
            ok = ok and bool((gm[lo:hi] == em).all() and (gc[lo:hi] == ec).all())
        out["parity_sample_ok"] = ok
        out["parity_sample"] = "rank 0's reads [0,2000), middle 2000, last 2000 vs oracle/movi_oracle.c on the blocked-thresholds image: matched lengths and counts equal"
        cpu.close()
    index.close()
    return out if rank == 0 else None


def long_reads_leg(torch, dist, world, rank, dev, stream, index, idx_dir, rows, row_bytes, n3, with_few, PG):
    """c3 on the resident index: n3 x 10 kbp reads per GPU (8 % substitutions; every rank draws its own from the text), plain PML
    and -- BASELINE config 3 is "PML + --classify" -- with the classification bins fused into the walk (vector + bins, bins
    only), each with its roofline object.  Returns the `long_reads` (and `few_long_reads`) objects on rank 0."""
    import subprocess
    w3 = WORKLOADS["c3"]
    L3 = w3["read_len"]
    tool = os.path.join(ROOT, "tools", "build_index")
    text = os.path.join(idx_dir, "text.bin")
    if rank == 0 and not os.path.exists(text):
        subprocess.check_call([tool, "pangenome", str(PG["anc"]), str(PG["genomes"]), str(PG["snp"]), str(PG["seed"]), "6",
                               idx_dir, "text-only"], stderr=subprocess.DEVNULL)
    if world > 1:
        dist.barrier()
    rf = os.path.join(idx_dir, "reads_%dx%d_%g%s.bin" % (n3, L3, w3["sub"], "" if rank == 0 else "_rank%d" % rank))
    if not os.path.exists(rf):
        subprocess.check_call([tool, "reads", text, str(n3), str(L3), str(w3["sub"]), str(PG["seed"] + 1000 * rank), rf + ".tmp%d" % rank])
        os.rename(rf + ".tmp%d" % rank, rf)
    b3 = torch.from_numpy(np.fromfile(rf, np.uint8, count=n3 * L3)).to(dev)
    if rank != 0:
        os.remove(rf)
    o3 = torch.from_numpy((np.arange(n3 + 1, dtype=np.uint64) * np.uint64(L3)).view(np.int64)).to(dev)
    out3 = torch.empty(n3 * L3, dtype=torch.int16, device=dev)
    err3 = torch.zeros(n3, dtype=torch.uint8, device=dev)
    d_a = torch.zeros(n3, dtype=torch.int32, device=dev)
    d_b = torch.zeros(n3, dtype=torch.int32, device=dev)
    d_s = torch.zeros(n3, dtype=torch.int64, device=dev)
    total = sum_over_ranks(torch, dist, world, dev, n3 * L3)
    k3 = 3
    res, verdicts = {}, {}
    for cm in (0, 1, 2):
        if cm == 0:
            run3 = lambda: index.pml_device(b3.data_ptr(), o3.data_ptr(), n3, n3 * L3, out3.data_ptr(), err3.data_ptr(), stream.cuda_stream, 0)
        else:
            run3 = lambda cm=cm: index.pml_classify_device(b3.data_ptr(), o3.data_ptr(), n3, n3 * L3, 150, 8, out3.data_ptr() if cm == 1 else 0,
                                                           d_a.data_ptr(), d_b.data_ptr(), d_s.data_ptr(), err3.data_ptr(), stream.cuda_stream, 0)
        run3()
        torch.cuda.synchronize()
        st3 = index.last_stats(stream.cuda_stream)
        dt3, per3, kern3 = timed_steps(torch, dist, world, dev, stream, run3, k3)
        launch = index.last_launch()
        leg = {"value": total * k3 / dt3 / 1e9, "unit": "Gbases/s", "steps": k3, "ms_per_step": dt3 / k3 * 1e3,
               "rank_seconds": [round(x, 4) for x in per3], "fused_classify": cm,
               "simt_efficiency": round(st3.lane_steps / (64.0 * st3.wave_steps), 4) if st3.wave_steps else None,
               "iterations_per_base": round(st3.lane_steps / (n3 * L3), 4) if st3.wave_steps else None,
               "fast_forwards_per_base": round(st3.fast_forwards / (n3 * L3), 4), "scans_per_base": round(st3.scans / (n3 * L3), 4),
               "errors": int(st3.errors), "segments": int(st3.segments), "rewalked_reads": int(st3.rewalked),
               "roofline": pml_roofline(table_bytes_walked(rows, row_bytes, launch), row_bytes, st3, n3 * L3, kern3, launch,
                                        *lookup_traffic("c3" if cm == 0 else "c3_classify%d" % cm, rows, n3, L3, launch["kernel"]))}
        if cm:
            verdicts[cm] = (d_a.clone(), d_b.clone(), d_s.clone())
        res[cm] = leg
    lr = dict(res[0])
    lr.update({"workload": "c3", "description": w3["desc"], "n_gpus": world, "reads_per_gpu": n3, "read_len": L3,
               "kernel": res[0]["roofline"]["kernel"], "launch": res[0]["roofline"]["launch"],
               "classify_vector_and_bins": res[1], "classify_bins_only": res[2],
               "classify_bins_agree": bool(all(torch.equal(x, y) for x, y in zip(verdicts[1], verdicts[2])))})
    if getattr(index, "dry", False):
        lr["dry_run"] = True
    few = None
    if with_few and world == 1:
        # the same reads, a quarter of them: too few walks to fill the GPU with one lane per read -- the shape the
        # segment-parallel walk is for; both ways, same launch otherwise
        n4 = n3 // 4
        few = {"workload": "first %d of the c3 reads" % n4, "unit": "Gbases/s"}
        for name, sl in (("one_lane_per_read", 0), ("segment_parallel", 2048)):
            index.set_option("seg_len", sl)
            run4 = lambda: index.pml_device(b3.data_ptr(), o3.data_ptr(), n4, n4 * L3, out3.data_ptr(), err3.data_ptr(),
                                            stream.cuda_stream, 0)
            run4()
            torch.cuda.synchronize()
            st4 = index.last_stats(stream.cuda_stream)
            t0 = time.perf_counter()
            for _ in range(k3):
                run4()
            torch.cuda.synchronize()
            few[name] = round(n4 * L3 * k3 / (time.perf_counter() - t0) / 1e9, 2)
            few[name + "_segments"] = int(st4.segments)
            if sl == 0:
                keep4 = out3[: n4 * L3].clone()
            else:
                few["identical"] = bool(torch.equal(keep4, out3[: n4 * L3]))
                del keep4
        index.set_option("seg_len", 2048)
    return (lr, few) if rank == 0 else (None, None)


def write_fasta(path, reads2d):
    """reads2d: uint8 [n, L] -> FASTA with fixed-width ids (>r0000000), written in one piece."""
    n, L = reads2d.shape
    rec = np.empty((n, 10 + L + 1), np.uint8)
    rec[:, 0] = ord(">")
    rec[:, 1] = ord("r")
    idx = np.arange(n)
    for k in range(7):
        rec[:, 8 - k] = ord("0") + (idx // 10 ** k) % 10
    rec[:, 9] = ord("\n")
    rec[:, 10: 10 + L] = reads2d
    rec[:, 10 + L] = ord("\n")
    rec.tofile(path)


def cli_measure(idx_dir, fa, out_prefix, modes=("no_output", "bpf"), extra=(), env=None, runs=2):
    """`movi query -i idx_dir -r fa --verbose` per output mode, best of `runs`: {mode: {value, seconds, stage_s, wall_s}}."""
    import glob
    import re
    import subprocess
    exe = os.path.join(ROOT, "movi_amd", "bin", "movi")
    res = {}
    for mode in modes:
        flags = ["--no-output"] if mode == "no_output" else ["-o", out_prefix]
        best, wall = None, None
        for _ in range(runs):
            for old in glob.glob(out_prefix + "*.bpf"):       # a fresh output file every run: ext4 flushes a truncated-and-rewritten file in close()
                os.remove(old)                                # (+30 ms inside the command's clock; profiles/r05_cli_path.txt)
            t0 = time.perf_counter()
            r = subprocess.run([exe, "query", "-i", idx_dir, "-r", fa, "--verbose"] + flags + list(extra), capture_output=True, timeout=900,
                               env=dict(os.environ, **env) if env else None)
            w = time.perf_counter() - t0
            if r.returncode != 0:
                raise RuntimeError(r.stderr.decode()[-300:])
            m = re.search(r"processing the reads: ([0-9.e+-]+) s \((\d+) bases; GPU calls ([0-9.e+-]+) s", r.stderr.decode())
            sec, nb = float(m.group(1)), int(m.group(2))
            if best is None or sec < best[0]:
                st = re.search(r"Stage times: parse ([0-9.e+-]+) s, GPU calls ([0-9.e+-]+) s, order \+ write ([0-9.e+-]+) s", r.stderr.decode())
                best = (sec, nb, float(m.group(3)), st.groups() if st else None)
                wall = w
        res[mode] = {"value": round(best[1] / best[0] / 1e9, 3), "seconds": round(best[0], 4), "gpu_calls_s": round(best[2], 4),
                     "stage_s": {"parse": float(best[3][0]), "gpu": float(best[3][1]), "write": float(best[3][2])} if best[3] else None,
                     "wall_s": round(wall, 3)}
    return res


def cli_path_leg(idx_dir, reads_150, reads_10k):
    """The drop-in a user touches: the `movi query` binary end to end on FASTA files of the same reads (page cache warm),
    --no-output and with the BPF file; rates exclude process start and index load (the command's own "processing the
    reads" clock, src/movi.cpp:387-389 prints the same), `wall_s` includes them.  Round 5: also `--gpus 2` with both logical
    GPUs on the box's one device (MOVI_SHARE_GPU=1: N independent loads, the reads sharded by bases over two host threads) so
    that the sharding code is timed at least once."""
    import tempfile
    out = {"unit": "Gbases/s", "note": "movi query on FASTA input, best of 2 runs; value = bases / the command's own read-processing time "
                                       "(parse + GPU calls + order + write, pipelined).  Since round 6 the parser's warm-up no longer scans the input "
                                       "while the index loads (round 5's did, outside this clock: its 10.2 / 4.8 Gbases/s excluded ~5 ms of the parse): "
                                       "every read of the input is inside the clock; wall_s adds process start, HIP init and index load",
           "input_read_inside_clock": True}
    tmp = tempfile.mkdtemp(prefix="movi_cli_")
    try:
        for name, (arr, L) in (("short_1Mx150", reads_150), ("long_100kx10k", reads_10k)):
            if arr is None:
                continue
            fa = os.path.join(tmp, name + ".fa")
            write_fasta(fa, arr.reshape(-1, L))
            for mode, v in cli_measure(idx_dir, fa, os.path.join(tmp, name)).items():
                out[name + "_" + mode] = v
            if name == "short_1Mx150":
                try:
                    out[name + "_no_output_gpus2_one_device"] = cli_measure(idx_dir, fa, os.path.join(tmp, name), modes=("no_output",), extra=("--gpus", "2"),
                                                                            env={"MOVI_SHARE_GPU": "1"})["no_output"]
                except Exception as e:                    # noqa: BLE001
                    out[name + "_no_output_gpus2_one_device"] = {"error": repr(e)[:200]}
            os.remove(fa)
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def usable_cores():
    """Host threads this process can really run: the CPU affinity mask capped by the cgroup CPU quota (the GPU boxes
    show 256 logical CPUs but grant 16 CPUs of quota: 256 OpenMP threads then run at 0.15-0.2 Gbases/s, 16 at 0.39;
    tests/studies/cpu_threads_sweep.py, profiles/r02_cpu_port_thread_sweep.txt)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                  # cgroup v2
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())             # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


SEED = 20260529
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
INFINITY_CACHE_BYTES = 256 << 20   # MI355X_MICROARCH.md: 256 MiB Infinity Cache (MALL) in front of HBM


# ---- `--dry-run`: the whole N-rank flow of this file on the CPU, over gloo, WITHOUT a GPU call (round 6) -------------------------
# No 8-GPU node has ever run this build, so everything around the kernels -- the self-spawn under torch.distributed.run, rank 0's
# synthesis of the tables, the one broadcast of the rows, the per-rank read hand-over, the barrier / max-over-ranks timing and the
# assembly of rank 0's JSON line -- is exercised here with the engine replaced by a stand-in whose calls do nothing (and say so:
# "dry_run": true, value = null).  tests/test_dist_cpu.py runs `--gpus 8 --dry-run` and holds its peak resident memory and wall time
# to the driver's limits.  What it cannot cover is RCCL itself (`dist.broadcast` on the nccl backend, ncclCommInitAll).
class _DryStats:
    bases = fast_forwards = scans = repositions = errors = lane_steps = wave_steps = segments = rewalked = 0


class DryIndex:
    """Stand-in for movi_amd.MoveIndex: same calls, no device, no results."""
    PREPARE_PML, PREPARE_COUNT, PREPARE_ZML = 1, 2, 4
    dry = True
    _h = None

    def __init__(self, rows_bytes=0):
        self.rows_bytes = rows_bytes

    @classmethod
    def from_device_rows(cls, cdesc, d_rows_ptr, device=0, keepalive=None):
        ix = cls(int(keepalive.numel()) if keepalive is not None else 0)
        ix._keep = keepalive
        return ix

    def set_option(self, key, value): pass
    def prepare(self, what=7, stream=0): return 0
    def pml_device(self, *a, **k): pass
    def pml_mask_device(self, *a, **k): pass
    def pml_expand_device(self, *a, **k): pass
    def pml_classify_device(self, *a, **k): pass
    def zml_device(self, *a, **k): pass
    def count_device(self, *a, **k): pass
    def last_stats(self, stream=0): return _DryStats()
    def last_launch(self): return {"kernel": "(dry run: no kernel was launched)", "variant": -1, "block_threads": 0, "waves_per_cu": 0, "segmented": 0, "idx64": 0, "staged": 0, "ahead": 0}
    def info(self, key): return 0.0
    def close(self): pass


class _DryEvent:
    def __init__(self, enable_timing=True): self.t = 0.0
    def record(self, stream=None): self.t = time.perf_counter()
    def elapsed_time(self, other): return (other.t - self.t) * 1e3 + 1e-6


class _DryStream:
    cuda_stream = 0


def install_dry_run(torch, movi_amd):
    """Replace what touches a device: torch.cuda.* used by this file, and the engine's handle class.  Returns the engine stand-in."""
    import types
    torch.cuda.is_available = lambda: True
    torch.cuda.set_device = lambda d: None
    torch.cuda.synchronize = lambda *a: None
    torch.cuda.empty_cache = lambda: None
    torch.cuda.current_stream = lambda *a: _DryStream()
    torch.cuda.Event = _DryEvent
    shim = types.SimpleNamespace(MoveIndex=DryIndex, parse_index_image=movi_amd.parse_index_image,
                                 pinned_empty=lambda n, dt: np.empty(n, dt), MoviError=movi_amd.MoviError)
    return shim


def peak_rss_mb():
    import resource
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0


def spawn_ranks(n):
    """`python bench.py --gpus N` started by hand (no torchrun): this process -- which has not touched the GPU, nor even
    imported torch -- starts N ranks under torch.distributed.run as a CHILD and relays its exit code; rank 0's JSON
    line goes to the inherited stdout.  (Never an exec: a process that has initialised the GPU must not be replaced.)"""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--read-len", type=int, default=0)
    ap.add_argument("--sub-rate", type=float, default=-1.0, help="substitution rate of the synthetic reads (synth workloads)")
    ap.add_argument("--classify", type=int, default=0, choices=[0, 1, 2],
                    help="PML with Classifier::classify bins fused into the walk (BASELINE config 3 is 'PML + --classify'): "
                         "1 = PML vectors + bins, 2 = bins only (--classify --filter: no PML vector is written)")
    ap.add_argument("--variant", type=int, default=-1, help="pml kernel variant (A/B measurement)")
    ap.add_argument("--zml-variant", type=int, default=-1, help="ZML kernel: 0 base-synchronous, 1 lane state machine (A/B)")
    ap.add_argument("--seg-len", type=int, default=-1, help="PML: segment length of the segment-parallel long-read path "
                    "(-1 = the engine's default of 2048, 0 = off)")
    ap.add_argument("--seg-probe", type=int, default=-1, help="segment-parallel policy (A/B): 0 = cut every eligible batch, 1 = probe (default)")
    ap.add_argument("--kmer-k", type=int, default=-1, help="top-of-walk table: first K bases of every read by one lookup (A/B; -1 = the engine's default)")
    ap.add_argument("--ftab-k", type=int, default=-1, help="count query's interval table: first K bases of the backward search by one lookup (A/B; -1 = the engine's default)")
    ap.add_argument("--stage-reads", type=int, default=-1, help="reads staged through LDS for short-read wavefronts (A/B: 0 off, 1 on; -1 = default)")
    ap.add_argument("--ahead-rows", type=int, default=-1, help="look-ahead rows: the table's second copy that resolves two bases per gather (A/B: 0 off, 1 on; -1 = the engine's default: on for tables whose copy fits the Infinity Cache)")
    ap.add_argument("--block-threads", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="movi_set_option(KEY, VALUE) on the handle before the timed region (A/B sweeps; repeatable)")
    ap.add_argument("--waves-per-cu", type=int, default=-1)
    ap.add_argument("--ragged", type=int, default=0, help="1: log-normal read lengths (mean = read_len); "
                    "2: same, lanes handed out longest-first (d_read_order)")
    ap.add_argument("--from-dir", default="", help="use DIR/index.movi + DIR/reads.bin (fixed-length reads, "
                    "--read-len) written by tools/build_index instead of the synthetic table")
    ap.add_argument("--reads-file", default="reads.bin")
    ap.add_argument("--query", default="pml", choices=["pml", "count", "zml"], help="count = backward-search count query "
                    "(BASELINE config 5 path); the headline metric is pml")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-long-reads", action="store_true", help="skip the secondary 100k x 10kbp measurement that the default "
                    "run (c2, N=1) appends as `long_reads` (north_star: 150 bp and 10 kbp reads)")
    ap.add_argument("--cpu-sample-reads", type=int, default=0)
    ap.add_argument("--no-big-table", action="store_true", help="skip the 1 B-row (8 GB, HBM-resident) leg that the default run "
                    "(c2, N=1) appends as `big_table` (BASELINE config 4's per-GPU shard, oracle-checked)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 5 s of back-to-back steps appended as `sustained`")
    ap.add_argument("--big-rows", type=int, default=1_000_000_000, help="rows of the `big_table` leg's table (tests shrink it)")
    ap.add_argument("--long-reads", type=int, default=0, help="reads per GPU of the `long_reads` leg (default: c3's 100 000; tests shrink it)")
    ap.add_argument("--quick", action="store_true", help="the timed region only: no cpu_baseline, long_reads, host_path, sustained, big_table (A/B sweeps)")
    ap.add_argument("--with-mask-leg", action="store_true", help="with --quick: keep the `mask_path` leg (the mask walk and the expand kernel timed on their own)")
    ap.add_argument("--dry-run", action="store_true", help="the N-rank flow on the CPU over gloo with a stand-in engine: no GPU call, no result (value null); "
                    "spawn, rank 0's synthesis, the broadcast, the read hand-over, timing and JSON assembly are real (tests/test_dist_cpu.py)")
    args = ap.parse_args()
    t_wall0 = time.time()
    if args.dry_run:
        args.no_cpu_baseline = args.no_sustained = True
    args.quick_only_walk = bool(args.quick)
    if args.quick:
        args.no_cpu_baseline = args.no_long_reads = args.no_big_table = args.no_sustained = True
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    import movi_amd
    from movi_amd._lib import IndexDescC
    from tools import synth

    if args.dry_run:
        movi_amd = install_dry_run(torch, movi_amd)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # test hook for 1-GPU boxes: MOVI_BENCH_SHARE_GPU=1 runs every rank on cuda:0 over gloo so that the
    # N>1 code path (index broadcast, per-rank read shards, max-over-ranks timing) can be exercised
    share_gpu = os.environ.get("MOVI_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cpu") if args.dry_run else torch.device("cuda", local_rank)
    rccl_ranks = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu or args.dry_run:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)          # backend "nccl" IS RCCL on ROCm
            rccl_ranks = dist.get_world_size()

    wl = dict(WORKLOADS[args.workload])
    if args.rows: wl["rows"] = args.rows
    if args.reads: wl["reads"] = args.reads
    if args.read_len: wl["read_len"] = args.read_len
    if args.sub_rate >= 0: wl["sub"] = args.sub_rate
    ROW_BYTES = {6: 8, 8: 8, 7: 8}       # resident row bytes: blocked- / sampled-thresholds rows are expanded to the mode-6 layout at upload
    mode, row_bytes = wl["mode"], ROW_BYTES[wl["mode"]]

    # ---- index: every rank derives the same host-side structure from the seed (needed to
    # draw reads); the DEVICE row table comes from rank 0 through one RCCL broadcast.
    t0 = time.time()
    six, file_img, reads_path = None, None, None
    barrier = (lambda: dist.barrier()) if world > 1 else (lambda: None)
    if args.from_dir:
        idx_dir, reads_path = args.from_dir, os.path.join(args.from_dir, args.reads_file)
        wl["desc"] = "index + reads from %s (tools/build_index: real BWT of a synthetic pangenome)" % args.from_dir
    elif wl["kind"] == "pangenome":
        # rank 0 builds / finds the cached index; if that is impossible on this box (no compiler, no
        # scratch space) every rank falls back to the random table and the JSON says so
        state = [None]
        if rank == 0:
            try:
                state[0] = ensure_pangenome(wl, world, rank, lambda: None)
            except Exception as e:                       # noqa: BLE001
                state[0] = "pangenome workload unavailable (%s): fell back to the random move table" % repr(e)[:200]
        if world > 1:
            dist.broadcast_object_list(state, src=0)
        if isinstance(state[0], str):
            wl.update(kind="synth", rows=10_000_000, desc=state[0])
        else:
            idx_dir, reads_path = state[0]
    if reads_path:
        file_img = np.fromfile(os.path.join(idx_dir, "index.movi"), np.uint8) if rank == 0 else None
        if rank == 0:
            fdesc = movi_amd.parse_index_image(file_img)[0]
            mode, row_bytes = fdesc.mode, fdesc.row_bytes
            wl["rows"], wl["mode"] = fdesc.r, fdesc.mode
    elif rank == 0:
        six = synth.synth_index(wl["rows"], mode=mode, seed=SEED)      # rank 0 only: the other ranks get the rows by broadcast, their reads from rank 0
    t_index_gen = time.time() - t0
    t0 = time.time()
    meta, d_rows = None, None
    if rank == 0:
        img = file_img if file_img is not None else six.image()
        _, cdesc, roff, rbytes = movi_amd.parse_index_image(img)
        id_blocks = (np.ctypeslib.as_array(C.cast(cdesc.id_blocks, C.POINTER(C.c_uint32)),
                                           shape=(int(cdesc.n_blocks) * int(cdesc.alphabet_size),)).copy() if mode == 8 else None)
        tally = (np.ctypeslib.as_array(C.cast(cdesc.tally_ids, C.POINTER(C.c_uint8)),
                                       shape=(int(cdesc.n_tally) * int(cdesc.alphabet_size) * 5,)).copy() if mode == 7 else None)
        meta = {"cdesc": bytes(cdesc), "id_blocks": id_blocks, "tally": tally}
        d_rows = torch.from_numpy(img[roff: roff + rbytes]).to(dev)     # rank 0 uploads the table once
    t_bcast = 0.0
    if world > 1:
        from movi_amd import dist as md
        torch.cuda.synchronize()
        tb = time.time()
        meta, d_rows = md.broadcast_index(meta, d_rows, src=0, device=dev)   # the one collective of this path
        torch.cuda.synchronize()
        t_bcast = time.time() - tb
    cdesc = IndexDescC.from_buffer_copy(meta["cdesc"])
    mode, row_bytes = int(cdesc.mode), ROW_BYTES[int(cdesc.mode)]
    wl["rows"], wl["mode"] = int(cdesc.r), mode
    id_blocks = meta["id_blocks"]
    cdesc.id_blocks = id_blocks.ctypes.data if id_blocks is not None else None
    tally = meta.get("tally")
    cdesc.tally_ids = tally.ctypes.data if tally is not None else None
    cdesc.separator_thresholds, cdesc.separator_map = None, None      # the bench workloads carry no separators
    cdesc.n_separator_thresholds = cdesc.n_separator_map = 0
    index = movi_amd.MoveIndex.from_device_rows(cdesc, d_rows.data_ptr(), device=local_rank, keepalive=d_rows)
    if args.variant >= 0:
        index.set_option("pml_variant", args.variant)
    if args.zml_variant >= 0:
        index.set_option("zml_variant", args.zml_variant)
    if args.seg_len >= 0:
        index.set_option("seg_len", args.seg_len)
    if args.seg_probe >= 0:
        index.set_option("seg_probe", args.seg_probe)
    if args.kmer_k >= 0:
        index.set_option("kmer_k", args.kmer_k)
    if args.stage_reads >= 0:
        index.set_option("stage_reads", args.stage_reads)
    if args.ahead_rows >= 0:
        index.set_option("ahead_rows", args.ahead_rows)
    if args.ftab_k >= 0:
        index.set_option("ftab_k", args.ftab_k)
    if args.block_threads:
        index.set_option("block_threads", args.block_threads)
    if args.waves_per_cu >= 0:
        index.set_option("waves_per_cu", args.waves_per_cu)
    for kv in args.opt:
        key, _, val = kv.partition("=")
        index.set_option(key, int(val))
    t_index_upload = time.time() - t0
    # the handle's derived tables (top-of-walk / interval table, look-ahead rows, checkpoints) are built here, by name, not inside
    # the first warm-up step (movi_index_prepare, round 5)
    t0 = time.time()
    derived_bytes = index.prepare({"pml": index.PREPARE_PML, "count": index.PREPARE_COUNT, "zml": index.PREPARE_ZML}[args.query])
    t_prepare = time.time() - t0

    # ---- reads: each rank draws its own shard (seed + rank)
    t0 = time.time()
    lens = None
    if reads_path:
        L = wl["read_len"]
        per = min(wl["reads"], os.path.getsize(reads_path) // L // world)
        wl["reads"] = per
        bases = np.fromfile(reads_path, np.uint8, count=per * L, offset=rank * per * L)
        offs = (np.arange(per + 1, dtype=np.uint64) * np.uint64(L))
    else:
        def lens_of(p):
            g = np.random.default_rng(SEED + 77 + p)
            return np.clip(g.lognormal(np.log(wl["read_len"]) - 0.08, 0.4, size=wl["reads"]), 20, 5 * wl["read_len"]).astype(np.uint64)
        bases, offs = draw_synth_reads(torch, dist, world, rank, dev, synth, six, wl["reads"], wl["read_len"], SEED + 1, wl["sub"],
                                       lens_of if args.ragged else None)
        if args.ragged:
            lens = (offs[1:] - offs[:-1]).astype(np.uint64)
    t_reads_gen = time.time() - t0
    n_reads, n_bases = wl["reads"], int(bases.size)
    d_bases = torch.from_numpy(bases).to(dev)
    d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
    d_out = torch.empty(n_bases, dtype=torch.int16, device=dev)
    d_err = torch.zeros(n_reads, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream()
    d_order = 0                                      # uniform read length: no length sort needed
    if args.ragged == 2:
        order_t = torch.from_numpy(np.argsort(-(lens.astype(np.int64)), kind="stable").astype(np.uint32).view(np.int32)).to(dev)
        d_order = order_t.data_ptr()

    if args.query == "count":
        d_matched = torch.zeros(n_reads, dtype=torch.int64, device=dev)
        d_count = torch.zeros(n_reads, dtype=torch.int64, device=dev)

    if args.classify:
        d_above = torch.zeros(n_reads, dtype=torch.int32, device=dev)
        d_below = torch.zeros(n_reads, dtype=torch.int32, device=dev)
        d_summax = torch.zeros(n_reads, dtype=torch.int64, device=dev)

    def step():
        if args.classify and args.query == "pml":
            index.pml_classify_device(d_bases.data_ptr(), d_offs.data_ptr(), n_reads, n_bases, 150, 8,
                                      d_out.data_ptr() if args.classify == 1 else 0, d_above.data_ptr(),
                                      d_below.data_ptr(), d_summax.data_ptr(), d_err.data_ptr(), stream.cuda_stream,
                                      d_order)
        elif args.query == "count":
            index.count_device(d_bases.data_ptr(), d_offs.data_ptr(), n_reads, n_bases, d_matched.data_ptr(),
                               d_count.data_ptr(), d_err.data_ptr(), stream.cuda_stream, d_order)
        elif args.query == "zml":
            index.zml_device(d_bases.data_ptr(), d_offs.data_ptr(), n_reads, n_bases, d_out.data_ptr(),
                             d_err.data_ptr(), stream.cuda_stream, d_order)
        else:
            index.pml_device(d_bases.data_ptr(), d_offs.data_ptr(), n_reads, n_bases, d_out.data_ptr(),
                             d_err.data_ptr(), stream.cuda_stream, d_order)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    st = index.last_stats(stream.cuda_stream)
    if st.errors:
        raise SystemExit("kernel flagged %d reads with invariant violations" % st.errors)

    # ---- timed region: exactly K steps, barrier + synchronize on both sides, max over ranks
    elapsed, rank_seconds, avg_kern_s = timed_steps(torch, dist, world, dev, stream, step, args.steps)
    total_bases_per_step = sum_over_ranks(torch, dist, world, dev, n_bases)

    # which kernel the library's launch policy picked for this batch, and with which shape: asked, not assumed
    launch = index.last_launch()
    f_bar = st.fast_forwards / max(n_bases, 1)
    s_bar = st.scans / max(n_bases, 1)
    bytes_per_base = row_bytes * (1.0 + f_bar + s_bar) + 1 + 2
    work_bases = n_bases
    if args.query == "count":
        # per base the search actually extends over: two LF walkers + interval-shrink rows + 1 base in
        work_bases = max(int(d_matched.sum().item()), 1)
        f_bar, s_bar = st.fast_forwards / work_bases / 2.0, st.scans / work_bases
        bytes_per_base = 2 * row_bytes * (1.0 + f_bar) + row_bytes * s_bar + 1
    if args.query == "zml":
        # every base: two LF walkers (fast-forwards counted over both) + interval-shrink rows + 1 base in + 2 out
        f_bar, s_bar = st.fast_forwards / max(n_bases, 1) / 2.0, st.scans / max(n_bases, 1)
        bytes_per_base = 2 * row_bytes * (1.0 + f_bar) + row_bytes * s_bar + 1 + 2
    achieved_gbs = bytes_per_base * work_bases / avg_kern_s / 1e9
    value = total_bases_per_step * args.steps / elapsed / 1e9       # Gbases/s, whole job
    # ---- >= 5 s of the same step back to back (default run only, after the timed region, never part of `value`): the
    # timed region above is K x 3 ms -- statistically fine (HIP events agree with rocprofv3 to 0.1 %) but too short for a
    # 1 Hz utilisation sampler to see the GPU busy at all; this leg is the sustained figure
    sustained = None
    default_run = (rank == 0 and world == 1 and args.workload == "c2" and args.query == "pml" and not args.classify
                   and args.variant < 0 and not args.from_dir and not args.dry_run)
    if default_run and not args.no_sustained:
        try:
            n_sus = max(args.steps, int(5.5 / max(avg_kern_s, 1e-6)))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_sus):
                step()
            torch.cuda.synchronize()
            dts = time.perf_counter() - t0
            sustained = {"steps": n_sus, "seconds": round(dts, 3), "value": n_bases * n_sus / dts / 1e9,
                                   "unit": "Gbases/s", "ms_per_step": dts / n_sus * 1e3}
        except Exception as e:                            # noqa: BLE001
            sustained = {"error": repr(e)[:200]}

    # ---- the same batch as RESET MASKS (round 6; default run only, never part of `value`): the walk that writes one bit per base
    # (movi_pml_mask_device), and pml_expand_kernel turning the words back into the u16 vector -- each timed with HIP events
    mask_path = None
    if default_run and (not args.quick_only_walk or args.with_mask_leg):
        try:
            from movi_amd.engine import mask_words
            d_words = torch.zeros(mask_words(n_reads, n_bases), dtype=torch.int32, device=dev)
            d_out2 = torch.empty(n_bases, dtype=torch.int16, device=dev)
            def mask_step():
                index.pml_mask_device(d_bases.data_ptr(), d_offs.data_ptr(), n_reads, n_bases, d_words.data_ptr(), 0, d_err.data_ptr(), stream.cuda_stream)
            def expand_step():
                index.pml_expand_device(d_words.data_ptr(), d_offs.data_ptr(), n_reads, n_bases, d_out2.data_ptr(), 0, stream.cuda_stream)
            for _ in range(2):
                mask_step(); expand_step()
            torch.cuda.synchronize()
            mlaunch = index.last_launch()
            _, _, mk_s = timed_steps(torch, dist, 1, dev, stream, mask_step, args.steps)
            _, _, ex_s = timed_steps(torch, dist, 1, dev, stream, expand_step, args.steps)
            same = bool(torch.equal(d_out2, d_out))
            mask_path = {"walk_kernel": mlaunch["kernel"], "walk_ms": mk_s * 1e3, "expand_ms": ex_s * 1e3,
                         "masks_gbases_s": n_bases / mk_s / 1e9, "masks_plus_expand_gbases_s": n_bases / (mk_s + ex_s) / 1e9,
                         "expand_write_gb_s": 2.0 * n_bases / ex_s / 1e9, "bytes_out_per_base": 4.0 * d_words.numel() / n_bases,
                         "expanded_equals_vector_walk": same}
            if not same:
                print("PARITY FAILURE: masks + expand differ from the vector walk", file=sys.stderr)
            del d_words, d_out2
            index.pml_device(d_bases.data_ptr(), d_offs.data_ptr(), n_reads, n_bases, d_out.data_ptr(), d_err.data_ptr(), stream.cuda_stream, d_order)
            torch.cuda.synchronize()
        except Exception as e:                            # noqa: BLE001
            mask_path = {"error": repr(e)[:200]}

    # ---- the ZML parse and the count query of the same batch on the same resident index (default run only, never part of `value`)
    zml_c2 = count_c2 = None
    if default_run and not args.quick_only_walk:
        try:
            zml_c2 = zml_leg(torch, dist, 1, dev, stream, index, wl["rows"], row_bytes, d_bases, d_offs, n_reads, n_bases, d_out, d_err, "c2_zml", wl["read_len"])
            d_m2 = torch.zeros(n_reads, dtype=torch.int64, device=dev)
            d_c2 = torch.zeros(n_reads, dtype=torch.int64, device=dev)
            index.prepare(index.PREPARE_COUNT)
            runc = lambda: index.count_device(d_bases.data_ptr(), d_offs.data_ptr(), n_reads, n_bases, d_m2.data_ptr(), d_c2.data_ptr(), d_err.data_ptr(), stream.cuda_stream, 0)
            runc()
            torch.cuda.synchronize()
            stc = index.last_stats(stream.cuda_stream)
            dtc, _, kern_c = timed_steps(torch, dist, 1, dev, stream, runc, 5)
            cl = index.last_launch()
            ctr, ctsrc = lookup_traffic("c2_count", wl["rows"], n_reads, wl["read_len"], cl["kernel"])
            count_c2 = {"value": n_bases * 5 / dtc / 1e9, "unit": "Gbases/s (read bases)", "steps": 5, "ms_per_step": dtc / 5 * 1e3, "kernel": cl["kernel"],
                        "matched_bases_per_read": round(float(d_m2.sum().item()) / n_reads, 2),
                        "roofline": count_roofline(table_bytes_walked(wl["rows"], row_bytes, cl), row_bytes, stc, int(d_m2.sum().item()), kern_c, cl, ctr, ctsrc)}
            del d_m2, d_c2
            index.pml_device(d_bases.data_ptr(), d_offs.data_ptr(), n_reads, n_bases, d_out.data_ptr(), d_err.data_ptr(), stream.cuda_stream, d_order)
            torch.cuda.synchronize()
        except Exception as e:                            # noqa: BLE001
            zml_c2 = {"error": repr(e)[:200]}

    tkey = args.workload + ("" if args.query == "pml" else "_" + args.query) + ("_classify%d" % args.classify if args.classify else "")
    traffic, traffic_src = lookup_traffic(tkey, wl["rows"], wl["reads"], wl["read_len"], launch["kernel"])

    result = {
        "metric": {"pml": "PML", "count": "count", "zml": "ZML"}[args.query] + " query Gbases/s on " +
                  {6: "regular-thresholds", 8: "blocked-thresholds", 7: "sampled-thresholds"}[mode] + " index",
        "value": value, "unit": "Gbases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "rank_seconds": [round(x, 4) for x in rank_seconds],
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": args.workload, "description": wl["desc"], "rows": wl["rows"], "mode": mode,
                   "reads_per_gpu": n_reads, "read_len": wl["read_len"], "bases_per_step_per_gpu": n_bases,
                   "seed": SEED, "parallelism": "reads sharded x%d, index replicated (1 RCCL broadcast)" % world,
                   "fast_forwards_per_base": round(f_bar, 4), "scans_per_base": round(s_bar, 4),
                   "reposition_frac": round(st.repositions / max(n_bases, 1), 4),
                   "simt_efficiency": round(st.lane_steps / (64.0 * st.wave_steps), 4) if st.wave_steps else None,
                   "iterations_per_base": round(st.lane_steps / max(n_bases, 1), 4) if st.wave_steps else None,
                   "algorithmic_bytes_per_base": round(bytes_per_base, 3),
                   "segments": int(st.segments), "rewalked_reads": int(st.rewalked), "seg_len": args.seg_len,
                   "query": args.query, "fused_classify": args.classify, "matched_bases_per_step": work_bases, "pml_variant": args.variant, "kmer_k": args.kmer_k, "ahead_rows": args.ahead_rows, "ftab_k": args.ftab_k, "ragged": args.ragged, "waves_per_cu": args.waves_per_cu, "block_threads": args.block_threads, "index_gen_s": round(t_index_gen, 2), "prepare_s": round(t_prepare, 3), "derived_bytes": derived_bytes,
                   "index_upload_s": round(t_index_upload, 2), "index_broadcast_s": round(t_bcast, 3),
                   "reads_gen_s": round(t_reads_gen, 2),
                   "no_ff_share": round(index.info("ahead_no_ff"), 4), "derived_table_bytes": int(index.info("derived_bytes"))},
        "rccl_ranks": rccl_ranks, "index_broadcast_s": round(t_bcast, 4),
        "index_broadcast_gb_s": round(int(d_rows.numel()) / t_bcast / 1e9, 2) if t_bcast > 0 else None,
        "index_broadcast_bytes": int(d_rows.numel()),
        # bound: by the bytes the row gathers walk -- the look-ahead copy is 16 B per row; the 256 MB top-of-walk / interval table,
        # one lookup per read (~0.1 B per base), is reported beside it ("side_table_bytes"), not folded into the label.  `achieved`
        # prices the REFERENCE's 8-byte rows, whichever layout the launch walked on.
        "roofline": {"bound": "hbm", "gathers_served_from": served_from(table_bytes_walked(wl["rows"], row_bytes, launch)),
                     "working_set_bytes": table_bytes_walked(wl["rows"], row_bytes, launch),
                     "side_table_bytes": SIDE_TABLE_BYTES if args.query != "zml" and args.kmer_k != 0 else 0,
                     "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": traffic_src,
                     "kernel": launch["kernel"], "launch": launch,
                     "kernel_ms_avg": avg_kern_s * 1e3,
                     # iterations of the lane automata (each one gather of a row window) and table rows the reference's walk reads
                     "lane_iterations_per_s": st.lane_steps / avg_kern_s if st.wave_steps else None,
                     "rows_read_per_s": ((1.0 + f_bar + s_bar) * n_bases if args.query == "pml"
                                         else (2.0 * (1.0 + f_bar) + s_bar) * work_bases) / avg_kern_s},
    }

    if sustained is not None:
        result["sustained"] = sustained
    if zml_c2 is not None:
        result["zml"] = zml_c2
    if count_c2 is not None:
        result["count"] = count_c2
    if mask_path is not None:
        result["mask_path"] = mask_path
        if mask_path.get("expanded_equals_vector_walk") is False:
            result["parity_sample_ok"] = False
    oracle_sample = None
    # ---- CPU baseline: the oracle restatement (scalar port, 16 strands/thread + prefetch,
    # OpenMP over read groups) on a bounded sample of the same reads, rank 0, N == 1 only.
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.query == "zml":
        from oracle.oracle import Oracle
        cores = usable_cores()
        cpu = Oracle(file_img if file_img is not None else six.image())
        sample = args.cpu_sample_reads or max(1, min(n_reads, int(15e6 * cores / 8) // max(wl["read_len"], 1)))
        sb = bases[: int(offs[sample])]
        so = offs[: sample + 1]
        passes, dt, exp = 0, 0.0, None
        while dt < 10.0 and passes < 64:
            t0 = time.perf_counter()
            exp = cpu.zml_batch(sb, so, threads=cores)
            dt += time.perf_counter() - t0
            passes += 1
        got = d_out[: sb.size].cpu().numpy().view(np.uint16)
        result["cpu_baseline"] = {"value": sb.size * passes / dt / 1e9, "unit": "Gbases/s", "cores": cores,
                                  "kind": "port",
                                  "sample": "first %d reads (%d bases) of the same batch x %d passes, "
                                            "oracle/movi_oracle.c oracle_zml_batch (scalar port of query_zml, "
                                            "%d OpenMP threads, no software prefetch), %.2f s"
                                            % (sample, sb.size, passes, cores, dt)}
        result["parity_sample_ok"] = bool((got == exp).all())
        if not result["parity_sample_ok"]:
            print("PARITY FAILURE on the cpu_baseline sample", file=sys.stderr)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.query == "pml" and args.classify != 2:
        from oracle.oracle import Oracle
        cores = usable_cores()
        cpu = Oracle(file_img if file_img is not None else six.image())
        sample = args.cpu_sample_reads or max(1, min(n_reads, int(30e6 * cores / 8) // max(wl["read_len"], 1)))
        sb = bases[: int(offs[sample])]
        so = offs[: sample + 1]
        cpu.pml_batch(sb[: int(so[min(sample, 64)])], so[: min(sample, 64) + 1], threads=cores)   # warm
        passes, dt, exp = 0, 0.0, None
        while dt < 10.0 and passes < 64:                # ~10 s of CPU work, bounded
            t0 = time.perf_counter()
            exp, _, _ = cpu.pml_batch(sb, so, threads=cores, strands=16)
            dt += time.perf_counter() - t0
            passes += 1
        got = d_out[: sb.size].cpu().numpy().view(np.uint16)
        result["cpu_baseline"] = {"value": sb.size * passes / dt / 1e9, "unit": "Gbases/s", "cores": cores,
                                  "kind": "port",
                                  "sample": "first %d reads (%d bases) of the same batch x %d passes, "
                                            "oracle/movi_oracle.c oracle_pml_batch (scalar port of the reference's "
                                            "strand scheduler: %d OpenMP threads x 16 strands + prefetch), %.2f s"
                                            % (sample, sb.size, passes, cores, dt)}
        result["parity_sample_ok"] = bool((got == exp).all())
        oracle_sample = exp
        if not result["parity_sample_ok"]:
            print("PARITY FAILURE on the cpu_baseline sample", file=sys.stderr)
    # ---- secondary figure, default run (any N, after the timed region, never part of `value`): BASELINE config 3,
    # 100 k x 10 kbp reads per GPU on the same resident index, plain and with --classify fused into the walk
    legs_run = (args.workload in ("c2", "tinypg") and args.query == "pml" and not args.classify and not args.from_dir and args.variant < 0)
    if legs_run and not args.no_long_reads and reads_path:
        try:
            del d_out, d_bases
            torch.cuda.empty_cache()
            lr, few = long_reads_leg(torch, dist, world, rank, dev, stream, index, idx_dir, wl["rows"], row_bytes,
                                     args.long_reads or WORKLOADS["c3"]["reads"], with_few=not args.long_reads, PG=pg_of(wl))
            if rank == 0:
                result["long_reads"] = lr
                if few:
                    result["few_long_reads"] = few
        except Exception as e:                            # noqa: BLE001 -- never lose the headline line to the extra
            if world > 1:
                raise
            result["long_reads"] = {"error": repr(e)[:200]}
    # ---- PCIe-inclusive rate of the boundary's host entry point ("pre-parsed reads in pinned host memory to
    # PMLs in pinned host memory"), default run only, after the timed region, never part of `value`: the same batch
    # through movi_pml_host from pageable buffers (synchronous path) and from page-locked ones (overlapped path)
    if (rank == 0 and world == 1 and args.workload in ("c2", "tinypg") and args.query == "pml" and not args.classify
            and args.variant < 0 and not args.no_cpu_baseline):
        try:
            from movi_amd._lib import QueryStatsC, check, lib
            hp = {"unit": "Gbases/s", "note": "movi_pml_host, host buffers in and out, best of 3 calls after a warm-up call.  Round 6: by default only "
                                              "RESET MASKS cross PCIe on the way back (1 B up + 1/8 B down per base) and the u16 vector is expanded into the caller's "
                                              "buffer by host worker threads beside the walks (`host_masks` -1 = 1 for calls of >= 2^22 bases): a chunk's words go to the "
                                              "pool by a host function on its stream, sixteen chunks per call.  *_vector = `host_masks` 0, the vector itself comes "
                                              "down (1 B up + 2 B down: rounds 1-5), *_mixed = 2 (both ways side by side, `host_mask_share` 70 % as masks); masks_only = "
                                              "movi_pml_mask_host (the masks are the result).  pageable_synchronous = "
                                              "host_autopin off (upload, walk, download one after the other), pageable = the default (a big call page-locks the caller's "
                                              "reads for its duration and overlaps), page_locked = caller-allocated page-locked buffers; all checked against each other "
                                              "and the cpu_baseline's oracle sample",
                  "host_threads": usable_cores()}
            h_offs = np.ascontiguousarray(offs, np.uint64)
            stq = QueryStatsC()
            torch.cuda.synchronize()
            ref_out = None
            for name, mk, via in (("pageable_synchronous", lambda n, dt: np.empty(n, dt), -1), ("pageable", lambda n, dt: np.empty(n, dt), -1),
                                  ("page_locked", movi_amd.pinned_empty, -1), ("page_locked_vector", movi_amd.pinned_empty, 0),
                                  ("page_locked_mixed", movi_amd.pinned_empty, 2), ("pageable_vector", lambda n, dt: np.empty(n, dt), 0), ("masks_only", movi_amd.pinned_empty, -1)):
                # pageable_synchronous: "host_autopin" 0 = upload, walk, download one after the other; pageable: the default --
                # a big call page-locks the caller's buffers for its duration and overlaps the three
                index.set_option("host_autopin", 0 if name == "pageable_synchronous" else 1)
                index.set_option("host_masks", via)
                hb = mk(n_bases, np.uint8)
                hb[:] = bases
                if name == "masks_only":
                    from movi_amd.engine import expand_masks_host, mask_words
                    ho = np.zeros(mask_words(n_reads, n_bases), np.uint32)
                    call = lambda: check(lib().movi_pml_mask_host(index._h, hb.ctypes.data, h_offs.ctypes.data, n_reads, ho.ctypes.data, None, C.byref(stq)))
                else:
                    ho = mk(n_bases, np.uint16)
                    ho[:] = 0xFFFF
                    call = lambda: check(lib().movi_pml_host(index._h, hb.ctypes.data, h_offs.ctypes.data, n_reads, ho.ctypes.data, None, C.byref(stq)))
                ts = []
                for _ in range(4):
                    t0 = time.perf_counter()
                    call()
                    ts.append(time.perf_counter() - t0)
                if name == "masks_only":                   # (their expansion, timed on its own: what a consumer of the vector pays on the host)
                    t0 = time.perf_counter()
                    ho = expand_masks_host(ho, h_offs, threads=0)
                    hp["masks_only_host_expand_gbases_s"] = round(n_bases / (time.perf_counter() - t0) / 1e9, 2)
                if ref_out is None:
                    ref_out = ho.copy()
                hp[name] = round(n_bases / min(ts[1:]) / 1e9, 2)
                ok = bool((ho == ref_out).all())
                if oracle_sample is not None:
                    ok = ok and bool((ho[: oracle_sample.size] == oracle_sample).all())
                hp[name + "_ok"] = ok
                del hb, ho
            index.set_option("host_masks", -1)
            index.set_option("host_autopin", 1)
            result["host_path"] = hp
        except Exception as e:                            # noqa: BLE001 -- an extra, never worth the headline line
            result["host_path"] = {"error": repr(e)[:200]}
    default_run = (rank == 0 and world == 1 and args.workload in ("c2", "tinypg") and args.query == "pml" and not args.classify
                   and args.variant < 0 and not args.from_dir)
    # ---- the command-line drop-in end to end (default run only, after the timed region, never part of `value`)
    if default_run and not args.no_cpu_baseline and reads_path:
        try:
            w3 = WORKLOADS["c3"]
            rf = os.path.join(idx_dir, "reads_%dx%d_%g.bin" % (args.long_reads or w3["reads"], w3["read_len"], w3["sub"]))
            r10k = (np.fromfile(rf, np.uint8), w3["read_len"]) if os.path.exists(rf) else (None, 0)
            result["cli_path"] = cli_path_leg(idx_dir, (bases, wl["read_len"]), r10k)
        except Exception as e:                            # noqa: BLE001
            result["cli_path"] = {"error": repr(e)[:300]}
    # ---- BASELINE config 4's per-GPU shard (default run only, after the timed region, never part of `value`): the walk on
    # a 1 B-row / 8 GB table -- HBM-resident, 30-bit row ids, byte offsets beyond 2^32 -- with its own roofline, and three
    # slices of the batch (first / middle / last 2000 reads) compared with the oracle bit for bit, counters included
    if legs_run and not args.no_big_table:
        try:
            del index, d_rows
            torch.cuda.empty_cache()
            bt = big_table_leg(torch, dist, world, rank, dev, local_rank, stream, synth, movi_amd, usable_cores(), rows=args.big_rows)
            if rank == 0:
                result["big_table"] = bt
                if bt.get("parity_sample_ok") is False:
                    result["parity_sample_ok"] = False
                    print("PARITY FAILURE on the big_table oracle slices", file=sys.stderr)
        except Exception as e:                            # noqa: BLE001 -- an extra, never worth the headline line
            if world > 1:
                raise
            result["big_table"] = {"error": repr(e)[:300]}
    # every rank's peak resident memory and this process's wall time (the driver allows 1800 s per bench run)
    rss = [peak_rss_mb()]
    if world > 1:
        box = [None] * world
        dist.all_gather_object(box, rss[0])
        rss = box
    result["host_peak_rss_mb"] = [round(x, 1) for x in rss]
    result["wall_s"] = round(time.time() - t_wall0, 1)
    if args.dry_run:
        result["dry_run"] = True
        result["value_dry_run_meaningless"], result["value"] = result["value"], None
        result["data"] = "synthetic (DRY RUN: no GPU call was made, no result was computed)"
    parity_failed = rank == 0 and result.get("parity_sample_ok") is False
    if parity_failed:
        # a kernel that disagrees with the oracle has no throughput: the record keeps the measurement under
        # another name, `value` is nulled and the process fails
        result["value_unverified"], result["value"] = result["value"], None
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()
    if parity_failed:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
