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
