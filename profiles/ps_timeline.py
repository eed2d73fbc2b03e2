#!/usr/bin/env python3
"""Role timeline of the persistent decode kernel (WHISPER_HIP_PS_STAMPS=<file> python bench.py ...): per role kind and layer,
how long a role sat in its wait (prefetch issued -> producers arrived) and how long it ran after the wait; per step, the
critical chain (merge done -> next merge done).  Clock: s_memrealtime, 100 MHz (10 ns ticks).
    python profiles/ps_timeline.py stamps.bin [first_step last_step]"""
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint8)
hdr = raw[:16].view(np.int32)
n_steps, n_roles, grid, _ = [int(x) for x in hdr]
kinds = raw[16:16 + 4 * n_roles].view(np.int32)
st = raw[16 + 4 * n_roles:].view(np.uint64).reshape(n_steps, n_roles, 3).astype(np.int64)
names = {0: "attn", 1: "cross", 2: "mlp", 3: "logits", 4: "merge"}
ran = st[:, :, 2] > 0
last = int(np.nonzero(ran.any(1))[0].max()) + 1 if ran.any() else 0
lo = int(sys.argv[2]) if len(sys.argv) > 2 else min(8, last - 1)
hi = int(sys.argv[3]) if len(sys.argv) > 3 else last
print(f"steps stamped {last} of {n_steps}, roles/step {n_roles}, grid {grid}; statistics over steps [{lo}, {hi})")
tick = 0.01   # us
sel = st[lo:hi]
t0 = sel[:, :, 0].astype(float); t1 = sel[:, :, 1].astype(float); t2 = sel[:, :, 2].astype(float)
ok = sel[:, :, 2] > 0
print(f"{'role':<14}{'n':>5}{'wait us':>10}{'run us':>10}{'done-after-step-start':>24}")
merge_done = np.where(ok[:, kinds & 0xff == 4], t2[:, kinds & 0xff == 4], np.nan)
step_end = np.nanmax(merge_done, axis=1)
step_start = np.concatenate([[np.nan], step_end[:-1]])
for key in sorted(set(int(k) for k in kinds), key=lambda k: ((k & 0xff) >= 3, k >> 8, k & 0xff)):
    cols = kinds == key
    m = ok[:, cols]
    if not m.any():
        continue
    w = (t1[:, cols] - t0[:, cols])[m] * tick
    r = (t2[:, cols] - t1[:, cols])[m] * tick
    rel = (np.nanmax(np.where(m, t2[:, cols], np.nan), axis=1) - step_start) * tick
    nm = names[key & 0xff] + (f" L{key >> 8}" if (key & 0xff) < 3 else "")
    print(f"{nm:<14}{int(cols.sum()):>5}{w.mean():>10.2f}{r.mean():>10.2f}{np.nanmean(rel):>24.2f}")
d = np.diff(step_end) * tick
print(f"step time (merge done -> next merge done): mean {np.nanmean(d):.2f} us, median {np.nanmedian(d):.2f}, min {np.nanmin(d):.2f}, max {np.nanmax(d):.2f}")
