#!/usr/bin/env python3
"""One step of a kernel-trace database as a timeline: python scripts/rocprof_timeline.py <db> <marker kernel> [skip]
The step = the launches between the last two occurrences of <marker kernel> (e.g. opt_range_kernel), minus `skip` from the end.
Prints, per queue, start offset / duration / gap to the previous kernel of that queue, and the busy time of each queue."""
import sqlite3
import sys


def main() -> None:
    db, marker = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = list(c.execute(f"select name, start, end, {qcol or '0'} from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    a, b = marks[-2] + 1, marks[-1] + 1
    step = rows[a:b]
    t0 = step[0][1]
    last = {}
    busy = {}
    print(f"# step of {len(step)} launches, {(step[-1][2] - t0) / 1e3:.1f} us from first start to last end")
    for name, s, e, q in step:
        gap = (s - last[q]) / 1e3 if q in last else 0.0
        last[q] = e
        busy[q] = busy.get(q, 0.0) + (e - s) / 1e3
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
        print(f"q{q} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} gap {gap:7.1f}  {short}")
    for q, v in busy.items():
        print(f"# queue {q}: busy {v:.1f} us")


if __name__ == "__main__":
    main()
