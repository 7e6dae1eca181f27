#!/usr/bin/env python3
"""Timeline of one step from a rocprofv3 (rocpd sqlite) kernel trace: every kernel between two
consecutive launches of an anchor kernel (default: the first count_gemm_fwd), with start / end
relative to the anchor, duration, queue / stream id -- to see what overlaps what.
Usage: rocprof_timeline.py results.db [anchor substring] [which occurrence]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    anchor = sys.argv[2] if len(sys.argv) > 2 else "count_gemm_fwd"
    which = int(sys.argv[3]) if len(sys.argv) > 3 else -3
    cur = db.cursor()
    cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
    extra = [c for c in ("queue_id", "stream_id", "queue", "stream") if c in cols]
    rows = cur.execute("select name, start, end{} from kernels order by start".format(
        "".join(", " + c for c in extra))).fetchall()
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(idx) < 2:
        print("anchor not found; columns:", cols)
        return
    a = idx[which]
    b = idx[which + 1] if which + 1 < 0 or which + 1 < len(idx) else len(rows)
    t0 = rows[a][1]
    print("columns:", extra)
    prev_end = t0
    for r in rows[a:b]:
        name = re.sub(r"\(.*$", "", r[0]).replace("void ", "")[:70]
        print("{:9.1f} {:9.1f} {:8.1f}  {:<70} {}".format(
            (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, name,
            " ".join(str(v) for v in r[3:])))
    print("step span {:.1f} us".format((rows[b][1] - t0) / 1e3 if b < len(rows) else -1))


if __name__ == "__main__":
    main()
