#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs: per-kernel time stats and per-kernel PMC averages.

usage: prof_summary.py <dir-or-db> [...]   (prints a text table; --json for JSON)
"""
import glob
import json
import os
import sqlite3
import sys


def summarise(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    out = {"db": db, "kernels": [], "counters": []}
    try:
        for name, calls, total, avg, pct in cur.execute(
                "select name, total_calls, total_duration, average, percentage from top_kernels"):
            out["kernels"].append(dict(name=name, calls=calls, total_us=total, avg_us=avg, pct=pct))
    except sqlite3.Error:
        pass
    try:
        q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration), "
             "max(vgpr_count), max(sgpr_count), max(lds_block_size), max(grid_size), max(workgroup_size) "
             "from counters_collection group by kernel_name, counter_name")
        for row in cur.execute(q):
            out["counters"].append(dict(kernel=row[0], counter=row[1], n=row[2], avg=row[3], min=row[4], max=row[5],
                                        avg_duration_ns=row[6], vgpr=row[7], sgpr=row[8], lds=row[9],
                                        grid=row[10], wg=row[11]))
    except sqlite3.Error:
        pass
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    as_json = "--json" in sys.argv
    dbs = []
    for a in args:
        if os.path.isdir(a):
            dbs += sorted(glob.glob(os.path.join(a, "**", "*.db"), recursive=True))
        else:
            dbs.append(a)
    res = [summarise(d) for d in dbs]
    if as_json:
        print(json.dumps(res, indent=1))
        return
    for r in res:
        print("==", r["db"])
        for k in r["kernels"]:
            print("  KERNEL %-90s calls=%-4d avg=%10.2f us total=%12.2f us %6.2f%%"
                  % (k["name"][:90], k["calls"], k["avg_us"], k["total_us"], k["pct"]))
        for c in r["counters"]:
            if c["kernel"].startswith("void movi::") or "movi" in c["kernel"] or "chase" in c["kernel"]:
                print("  PMC %-60s %-22s n=%-3d avg=%.6g min=%.6g max=%.6g avg_dur=%.1f us vgpr=%s sgpr=%s lds=%s grid=%s wg=%s"
                      % (c["kernel"][:60], c["counter"], c["n"], c["avg"], c["min"], c["max"],
                         c["avg_duration_ns"] / 1e3, c["vgpr"], c["sgpr"], c["lds"], c["grid"], c["wg"]))


if __name__ == "__main__":
    main()
