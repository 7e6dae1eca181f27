#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per kernel name (and grid size)
calls, total, average duration.  Usage: rocprof_summary.py results.db [--filter scvae]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    return name[:90]


def main():
    path = sys.argv[1]
    flt = sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--filter" else None
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
    rows = cur.execute(
        "select name, grid_x, grid_y, grid_z, workgroup_x, (end - start) from kernels"
        if "grid_x" in cols else
        "select name, grid_size_x, grid_size_y, grid_size_z, workgroup_size_x, (end - start) "
        "from kernels").fetchall()
    agg = {}
    total = 0.0
    for name, gx, gy, gz, wx, dur in rows:
        if flt and flt not in name:
            continue
        key = (short(name), gx // max(wx, 1), gy, gz)
        a = agg.setdefault(key, [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
        total += dur
    print("{:<92} {:>14} {:>6} {:>11} {:>10} {:>10} {:>6}".format(
        "kernel", "grid(blocks)", "calls", "total_us", "avg_us", "min_us", "%"))
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("{:<92} {:>14} {:>6} {:>11.1f} {:>10.1f} {:>10.1f} {:>6.2f}".format(
            key[0], "{}x{}x{}".format(*key[1:]), a[0], a[1] / 1e3, a[1] / a[0] / 1e3,
            a[2] / 1e3, 100 * a[1] / total))


if __name__ == "__main__":
    main()
