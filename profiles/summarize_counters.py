#!/usr/bin/env python3
"""Per-kernel averages of the counters in a rocprofv3 --pmc database (one pass = one database):
    python profiles/summarize_counters.py <db> [<db> ...] > out.txt
Kernels are listed by total duration when the database also carries the kernel trace, else by name."""
import sqlite3
import sys


def tables(cur):
    return [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]


def main(paths):
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        names = tables(cur)
        view = "counters_collection" if "counters_collection" in names else None
        print(f"== {path}")
        if not view:
            print("   no counters_collection view; tables:", ", ".join(names[:20]))
            continue
        cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
        cname = "counter_name" if "counter_name" in cols else "name"
        rows = cur.execute(f"select kernel_name, {cname}, count(*), avg(value), sum(value) from {view} "
                           f"group by kernel_name, {cname} order by sum(value) desc").fetchall()
        for k, c, n, avg, tot in rows:
            print(f"{c:34s} n={n:6d} avg={avg:16.1f} sum={tot:18.1f}  {k[:110]}")


if __name__ == "__main__":
    main(sys.argv[1:])
