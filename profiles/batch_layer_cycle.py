#!/usr/bin/env python3
"""Batch-mode decode chain from a rocprofv3 --kernel-trace rocpd database: average duration of every kernel of the
per-layer cycle (the six skinny GEMMs apart: QKV, out, Wq, cross-out, lin1, lin2) and the idle gap that follows each.
    python profiles/batch_layer_cycle.py results.db"""
import sqlite3
import sys
from collections import defaultdict

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
names = ["QKV", "out-proj", "Wq", "cross-out", "lin1", "lin2"]
dur, gap = defaultdict(list), defaultdict(list)
n_sk = 0
for i, (n, s, e) in enumerate(rows):
    key = None
    if "dec_skinny_gemm" in n:
        key = "skinny " + names[n_sk % 6]
        n_sk += 1
    elif "dec_prepare" in n:
        n_sk = 0
        key = "dec_prepare"
    else:
        for k in ("dec_resolve_ln", "dec_self_attn", "dec_cross_attn_stream", "dec_gelu_fold", "dec_topk_rows", "gemm_f32"):
            if k in n:
                key = k
    if key is None:
        continue
    dur[key].append(e - s)
    if i + 1 < len(rows):
        g = rows[i + 1][1] - e
        if 0 < g < 100000:
            gap[key].append(g)
print(f"{'kernel':<28}{'n':>7}{'avg us':>9}{'min us':>9}{'gap after us':>14}")
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    d, g = dur[k], gap.get(k, [0])
    print(f"{k:<28}{len(d):>7}{sum(d) / len(d) / 1e3:>9.2f}{min(d) / 1e3:>9.2f}{sum(g) / max(1, len(g)) / 1e3:>14.2f}")
