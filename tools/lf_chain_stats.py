#!/usr/bin/env python3
"""LF-graph statistics of a regular-thresholds index (analysis tooling): how many rows are some row's
LF target, how many rows could have their target laid out right behind them (one predecessor per target: the one with the
longest run), and how often a walk arriving at the target needs no fast-forward -- all weighted by run length, i.e. by where
a walk that follows the text spends its steps.  usage: tools/lf_chain_stats.py INDEX_DIR/index.movi"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import movi_amd  # noqa: E402


def main(path):
    img = open(path, "rb").read()
    _, desc, off, _ = movi_amd.parse_index_image(img)
    if desc.mode != 6:
        raise SystemExit("regular-thresholds (mode 6) indexes only")
    r = desc.r
    rows = np.frombuffer(img, np.uint8, count=r * 8, offset=off).reshape(-1, 8)
    ids = rows[:, 0:4].copy().view(np.uint32).reshape(-1).astype(np.int64)
    n16 = rows[:, 4:6].copy().view(np.uint16).reshape(-1)
    o16 = rows[:, 6:8].copy().view(np.uint16).reshape(-1)
    ids |= ((o16 >> 12).astype(np.int64)) << 32
    n = (n16 & 0x7FF).astype(np.int64)
    offh = (o16 & 0x7FF).astype(np.int64)
    ok = ids < r
    ids = np.where(ok, ids, 0)
    indeg = np.bincount(ids[ok], minlength=r)
    print("rows %d, mean run length %.2f" % (r, n.mean()))
    print("rows that are some row's LF target: %.3f; in-degree histogram (0..5, 6+): %s"
          % ((indeg > 0).mean(), np.round(np.bincount(np.minimum(indeg, 6), minlength=7) / r, 3)))
    order = np.lexsort((-n, ids))                                  # per target: the predecessor with the longest run first
    first = np.ones(r, bool)
    first[1:] = ids[order][1:] != ids[order][:-1]
    adj = np.zeros(r, bool)
    adj[order[first]] = True
    adj &= ok
    print("rows whose target can be laid out right behind them: %.3f; weighted by run length: %.3f"
          % (adj.mean(), (n * adj).sum() / n.sum()))
    noff = np.clip(n[ids] - offh, 0, n) * ok                       # positions of the run that arrive below the target's length
    print("P(no fast-forward at the LF target), position-weighted: %.3f" % (noff.sum() / n.sum()))
    print("P(target adjacent AND no fast-forward): %.3f" % ((noff * adj).sum() / n.sum()))


if __name__ == "__main__":
    main(sys.argv[1])
