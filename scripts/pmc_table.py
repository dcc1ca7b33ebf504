#!/usr/bin/env python3
"""Per-kernel averages of every counter in one or more rocprofv3 PMC result databases:
    python scripts/pmc_table.py <results.db> [...]    ->  JSON {kernel: {counter: avg per launch, "launches": n}}"""
import json
import sqlite3
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


out = {}
for db in sys.argv[1:]:
    c = sqlite3.connect(db)
    for n, cn, k, v in c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        if n.startswith(("__amd", "at::", "void at::")):
            continue
        d = out.setdefault(short(n), {})
        d[cn] = v
        d["launches"] = k
print(json.dumps(out, indent=1, sort_keys=True))
