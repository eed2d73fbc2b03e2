#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace rocpd database (results.db) into a per-kernel stats CSV
(the same figures `rocprofv3 --stats` prints: calls, total, average, min, max, percentage)."""
import csv
import sqlite3
import sys


def main(db_path: str, out_csv: str) -> None:
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), int(r[4]), int(r[5]), round(100.0 * r[2] / tot, 3)])
    print(f"{len(rows)} kernels, {tot / 1e6:.2f} ms total -> {out_csv}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
