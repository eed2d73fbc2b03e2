#!/usr/bin/env python3
"""Role timeline of the persistent decode kernel (WHISPER_HIP_PS_STAMPS=<file> python bench.py ...).
Per role kind and layer, over the roles of LIVE rows only (a row whose window has ended skips its attention roles): the
time from role start to the wait being passed, the phases after the wait, and the arrive; per step, the critical chain.
Stamp slots: 0 role start, 1 wait passed, 2-5 phases inside the role, 6 done, 7 arrived.  Clock: 100 MHz (10 ns ticks).
    python profiles/ps_timeline.py stamps.bin [first_step last_step] [row]"""
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint8)
n_steps, n_roles, grid, ns = [int(x) for x in raw[:16].view(np.int32)]
kinds = raw[16:16 + 4 * n_roles].view(np.int32)
st = raw[16 + 4 * n_roles:].view(np.uint64).reshape(n_steps, n_roles, ns).astype(np.float64)
st[st == 0] = np.nan
names = {0: "attn", 1: "cross", 2: "mlp", 3: "logits", 4: "merge", 5: "finln"}
kind, layer, row = kinds & 0xff, (kinds >> 8) & 0xff, kinds >> 16
ran = ~np.isnan(st[:, :, 6])
last = int(np.nonzero(ran.any(1))[0].max()) + 1 if ran.any() else 0
lo = int(sys.argv[2]) if len(sys.argv) > 2 else min(8, max(last - 1, 0))
hi = int(sys.argv[3]) if len(sys.argv) > 3 else last
live_row = int(sys.argv[4]) if len(sys.argv) > 4 else 0
tick = 0.01
print(f"steps stamped {last} of {n_steps}, roles/step {n_roles}, grid {grid}; statistics over steps [{lo}, {hi}), attention roles of row {live_row}")
sel = st[lo:hi]
merge_done = sel[:, kind == 4, 7]
step_end = np.nanmax(merge_done, axis=1)
step_start = np.concatenate([[np.nan], step_end[:-1]])
hdr = f"{'role':<10}{'n':>4}{'wait':>8}{'w->p2':>8}{'p2->p3':>8}{'p3->p4':>8}{'p4->p5':>8}{'p5->done':>9}{'arrive':>8}{'run':>8}{'done@':>9}{'arrived@':>9}"
print(hdr)
for k in (0, 1, 2, 5, 3, 4):
    for l in sorted(set(layer[kind == k])):
        cols = (kind == k) & (layer == l)
        if k in (0, 1, 5):
            cols &= row == live_row
        if not cols.any():
            continue
        s = sel[:, cols, :]
        def d(a, b):
            x = (s[:, :, b] - s[:, :, a]) * tick
            return np.nanmean(x) if np.isfinite(x).any() else float("nan")
        rel_done = np.nanmean((np.nanmax(s[:, :, 6], axis=1) - step_start) * tick)
        rel_arr = np.nanmean((np.nanmax(s[:, :, 7], axis=1) - step_start) * tick)
        nm = names[k] + (f" L{l}" if k < 3 else "")
        print(f"{nm:<10}{int(cols.sum()):>4}{d(0,1):>8.2f}{d(1,2):>8.2f}{d(2,3):>8.2f}{d(3,4):>8.2f}{d(4,5):>8.2f}{d(5,6):>9.2f}{d(6,7):>8.2f}{d(1,6):>8.2f}{rel_done:>9.2f}{rel_arr:>9.2f}")
dd = np.diff(step_end) * tick
print(f"step time (merge arrived -> next merge arrived): mean {np.nanmean(dd):.2f} us, median {np.nanmedian(dd):.2f}, min {np.nanmin(dd):.2f}, max {np.nanmax(dd):.2f}")
