#!/usr/bin/env python3
"""Merge rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) of `bench.py` into
profiles/traffic.json, keyed by the bench configuration, per kernel and per launch.

    python scripts/make_traffic_json.py <key> <fetch.db> <write.db>

HBM bytes = 2 * FETCH_SIZE KiB (gfx950 counts 64 B per 128-B request on wide streaming reads,
MI355X_MICROARCH.md section HBM) + WRITE_SIZE KiB.
"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    return {short(n): (k, v) for n, k, v in c.execute(
        "select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name",
        (counter,))}


def main():
    key, fdb, wdb = sys.argv[1:4]
    f, w = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        if k.startswith(("__amd", "at::")):
            continue
        fk = f.get(k, (0, 0.0))[1]
        wk = w.get(k, (0, 0.0))[1]
        out[k] = {
            "FETCH_SIZE_KiB_avg": fk,
            "WRITE_SIZE_KiB_avg": wk,
            "read_bytes_per_launch_corrected": 2 * fk * 1024,
            "write_bytes_per_launch": wk * 1024,
            "hbm_bytes_per_launch": 2 * fk * 1024 + wk * 1024,
        }
    path = os.environ.get("TRAFFIC_OUT") or os.path.join(ROOT, "profiles", "traffic.json")
    allj = json.load(open(path)) if os.path.exists(path) else {}
    allj[key] = out
    json.dump(allj, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
