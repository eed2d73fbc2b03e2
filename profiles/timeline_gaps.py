#!/usr/bin/env python3
"""Decode-step timeline from a rocprofv3 --kernel-trace rocpd database: for the FASTEST bench step
(from its mel kernel to the next mel kernel), kernel busy time, idle gaps between consecutive kernels and
the per-kernel split -- shows whether the step is bound by kernel time or by dispatch gaps."""
import sqlite3
import sys
from collections import defaultdict


def main(db_path: str) -> None:
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    mel_idx = [i for i, r in enumerate(rows) if "mel_spectrogram" in r[0]] + [len(rows)]
    segs = [(a, b) for a, b in zip(mel_idx[:-1], mel_idx[1:]) if b - a > 50]      # whole steps (not the frontend leg)
    # the fastest whole step = a timed, graph-replayed one (bench.py's profiled passes launch eagerly and sync)
    a, b = min(segs, key=lambda ab: max(r[2] for r in rows[ab[0]:ab[1]]) - rows[ab[0]][1])
    seg = rows[a:b]
    t0, t1 = seg[0][1], max(r[2] for r in seg)
    busy = sum(r[2] - r[1] for r in seg)
    gaps = [seg[i + 1][1] - seg[i][2] for i in range(len(seg) - 1)]
    pos = [g for g in gaps if g > 0]
    print(f"kernels {len(seg)}  wall {(t1 - t0) / 1e3:.1f} us  busy {busy / 1e3:.1f} us  "
          f"idle {sum(pos) / 1e3:.1f} us  (mean gap {sum(pos) / max(1, len(pos)):.0f} ns, max {max(pos) / 1e3:.1f} us)")
    big = sorted(((g, i) for i, g in enumerate(gaps)), reverse=True)[:8]
    for g, i in big:
        print(f"  gap {g / 1e3:8.1f} us after {seg[i][0][:60]}")
    per = defaultdict(lambda: [0, 0])
    for n, s, e in seg:
        per[n][0] += 1
        per[n][1] += e - s
    for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {t / 1e3:9.1f} us  {c:5d} x {t / c / 1e3:6.2f} us  {n[:90]}")


if __name__ == "__main__":
    main(sys.argv[1])
