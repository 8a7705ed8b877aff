"""CPU suite: the N>1 path (index replication by one broadcast, reads sharded by bases,
results gathered in read order) with world_size 2 on the gloo backend."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import movi_amd
        from movi_amd import dist as md
        from oracle.oracle import Oracle
        from tools import synth
        bases = offs = None
        meta = rows = None
        if rank == 0:
            six = synth.synth_index(50000, mode=8, seed=3)
            img = six.image()
            desc, cdesc, roff, rbytes = movi_amd.parse_index_image(img)
            meta = {"image_head": img[:roff].tobytes(), "image_tail": img[roff + rbytes:].tobytes()}
            rows = torch.from_numpy(img[roff: roff + rbytes].copy())
            lens = np.random.default_rng(1).integers(1, 400, size=999)
            bases, offs = synth.synth_reads(six, 999, 0, seed=5, lens=lens)
        meta, rows = md.broadcast_index(meta, rows, src=0, device="cpu")
        # every rank can rebuild the same index image from what was broadcast
        img2 = meta["image_head"] + rows.numpy().tobytes() + meta["image_tail"]
        cpu = Oracle(img2)
        out = md.query_pml_sharded(lambda b, o: cpu.pml_batch(b, o, threads=1)[0], bases, offs, src=0)
        if rank == 0:
            exp, _, _ = cpu.pml_batch(bases, offs, threads=2)
            b = md.shard_bounds(offs, world)
            share = [(int(offs[b[p + 1]]) - int(offs[b[p]])) / int(offs[-1]) for p in range(world)]
            q.put(("ok", bool((out == exp).all()), share, int(rows.numel())))
        else:
            q.put(("ok", out is None, None, int(rows.numel())))
    except Exception as e:            # surface the failure in the parent
        q.put(("err", repr(e), None, None))
    finally:
        dist.destroy_process_group()


def test_two_rank_replicate_shard_gather(built_lib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[0] == "ok" for r in res), res
    assert all(r[1] for r in res)
    share = [r[2] for r in res if r[2] is not None][0]
    assert all(abs(s - 0.5) < 0.01 for s in share)             # balanced by bases
    assert res[0][3] == res[1][3] == 50000 * 6                 # the same table everywhere


def test_shard_bounds_properties():
    from movi_amd.dist import shard_bounds
    rng = np.random.default_rng(0)
    for _ in range(20):
        lens = rng.integers(0, 5000, size=int(rng.integers(1, 300)))
        offs = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
        for parts in (1, 2, 3, 8):
            b = shard_bounds(offs, parts)
            assert b[0] == 0 and b[-1] == len(lens) and all(x <= y for x, y in zip(b, b[1:]))
    assert shard_bounds(np.zeros(1, np.uint64), 4) == [0, 0, 0, 0, 0]     # empty batch


def test_bench_dry_run_eight_ranks(built_lib):
    """`bench.py --gpus 8 --dry-run`: the whole 8-rank flow of the bench -- self-spawn under torch.distributed.run, rank 0's synthesis,
    the index broadcast, the per-rank read hand-over, barrier / max-over-ranks timing, every leg's assembly into rank 0's ONE JSON
    line -- on the CPU over gloo with a stand-in engine (no GPU call, value null).  The first real 8-GPU run then only adds RCCL.
    Tables are shrunk (the container has 64 GB for 8 ranks); resident memory and wall time are held to what the driver allows."""
    import json
    import subprocess
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--workload", "tinypg", "--big-rows", "300000",
                        "--long-reads", "64", "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=1500, cwd=ROOT)
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                   # ONE JSON line, rank 0's
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["value"] is None and d["n_gpus"] == 8 and d["rccl_ranks"] == 0 and d["scaling"] == "weak"
    assert len(d["rank_seconds"]) == 8 and len(d["host_peak_rss_mb"]) == 8
    assert d["index_broadcast_s"] > 0 and d["index_broadcast_gb_s"] > 0 and d["index_broadcast_bytes"] == d["config"]["rows"] * 8
    for leg in ("long_reads", "big_table"):
        assert d[leg]["dry_run"] is True and d[leg]["n_gpus"] == 8 and len(d[leg]["rank_seconds"]) == 8, leg
    assert d["big_table"]["index_broadcast_gb_s"] > 0 and len(d["big_table"]["count"]["rank_seconds"]) == 8
    assert d["long_reads"]["classify_bins_agree"] is True
    # the driver's limits: 1800 s per bench run; rank 0 holds every rank's synthetic reads in turn, the others only their own shard
    assert wall < 1500 and d["wall_s"] < 1500
    assert max(d["host_peak_rss_mb"]) < 6000 and max(d["host_peak_rss_mb"][1:]) <= d["host_peak_rss_mb"][0] + 200
