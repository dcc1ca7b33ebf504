#!/usr/bin/env python3
"""Print the kernel timeline (start, duration, gap to the previous kernel) of a few steady-state steps
from a rocprofv3 --kernel-trace rocpd database:  python scripts/timeline.py <results.db> [anchor-substring] [nth]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2] if len(sys.argv) > 2 else "subtree_linear_kernel"
nth = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = list(c.execute("select name, start, end from kernels order by start"))
names = [r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0] for r in rows]
idx = [i for i, n in enumerate(names) if anchor in n]
i = idx[min(nth, len(idx) - 1)]
t0 = rows[max(i - 6, 1)][1]
for k in range(max(i - 6, 1), min(i + 12, len(rows))):
    print(f"{names[k][:48]:48s} start {(rows[k][1] - t0) / 1e3:8.1f} us  dur {(rows[k][2] - rows[k][1]) / 1e3:7.1f}  "
          f"gap_before {(rows[k][1] - rows[k - 1][2]) / 1e3:6.1f}")
