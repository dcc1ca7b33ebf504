#!/usr/bin/env python3
"""Turn rocprofv3's rocpd SQLite output into the text summaries committed under profiles/.

    python scripts/rocprof_summary.py <kernel-trace .db> [--pmc <counter .db> ...] > profiles/<name>.txt

Prints the `--stats`-style kernel table (calls, total, average, share) and, for each PMC database,
per-kernel averages of the collected counters.  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in
KiB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 counts 64 B per 128-B request on
wide coalesced streams, so the corrected read bytes are 2 x FETCH_SIZE x 1024.
"""
import argparse
import sqlite3


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("trace_db")
    ap.add_argument("--pmc", nargs="*", default=[])
    a = ap.parse_args()
    c = sqlite3.connect(a.trace_db)
    rows = list(c.execute(
        "select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3 "
        "from kernels group by name order by sum(duration) desc"))
    total = sum(r[2] for r in rows) or 1.0
    print(f"# kernel-trace stats from {a.trace_db} (durations in us)")
    print(f"{'kernel':60s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for n, k, tot, avg, mn, mx in rows:
        print(f"{short(n)[:60]:60s} {k:6d} {tot:12.1f} {avg:10.2f} {mn:10.2f} {mx:10.2f} {100*tot/total:6.2f}")
    print()
    print("# per (kernel, grid) averages")
    for n, gx, gy, k, avg in c.execute(
            "select name, grid_x, grid_y, count(*), avg(duration)/1e3 from kernels group by name, grid_x, grid_y "
            "order by name, grid_y desc, grid_x desc"):
        print(f"{short(n)[:60]:60s} grid=({gx},{gy}) calls={k} avg_us={avg:.2f}")
    for db in a.pmc:
        c2 = sqlite3.connect(db)
        print()
        print(f"# PMC averages per dispatch from {db}")
        for n, ctr, k, avg in c2.execute(
                "select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                "group by kernel_name, counter_name order by sum(value) desc"):
            extra = ""
            if ctr == "FETCH_SIZE":
                extra = f"  -> corrected read bytes/dispatch = {2*avg*1024/1e6:.2f} MB"
            if ctr == "WRITE_SIZE":
                extra = f"  -> write bytes/dispatch = {avg*1024/1e6:.2f} MB"
            print(f"{short(n)[:60]:60s} {ctr:12s} dispatches={k:5d} avg={avg:14.2f}{extra}")


if __name__ == "__main__":
    main()
