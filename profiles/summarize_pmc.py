#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, TCC
slot limits), corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950:
counter unit = KiB; FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read ->
read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE taken as is (uncalibrated)."""
import csv
import json
import sqlite3
import sys


def per_kernel(db_path):
    cur = sqlite3.connect(db_path).cursor()
    return {r[0]: (r[1], r[2]) for r in cur.execute(
        "select kernel_name, count(*), avg(value) from counters_collection group by kernel_name")}


def main(fetch_db, write_db, out_csv, out_json):
    f, w = per_kernel(fetch_db), per_kernel(write_db)
    rows = []
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, (0, 0))[1])):
        fk, wk = f.get(k, (0, 0.0)), w.get(k, (0, 0.0))
        rows.append({"kernel": k, "launches": fk[0] or wk[0], "FETCH_SIZE_KiB_avg": round(fk[1], 1),
                     "WRITE_SIZE_KiB_avg": round(wk[1], 1),
                     "hbm_bytes_per_launch_corrected": int(2 * fk[1] * 1024 + wk[1] * 1024)})
    with open(out_csv, "w", newline="") as fh:
        wr = csv.DictWriter(fh, fieldnames=list(rows[0]))
        wr.writeheader()
        wr.writerows(rows)
    json.dump({r["kernel"]: r["hbm_bytes_per_launch_corrected"] for r in rows}, open(out_json, "w"), indent=1)
    print(f"{len(rows)} kernels -> {out_csv}, {out_json}")


if __name__ == "__main__":
    main(*sys.argv[1:5])
