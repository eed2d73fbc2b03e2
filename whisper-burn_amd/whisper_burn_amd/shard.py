"""Multi-GPU long-audio transcription: shard the reference's windows over ranks, one
all-gather of token rows, host stitch.

Windows are independent in the reference -- the previous-token prompt is computed
(/root/reference/src/transcribe.rs:43-50) and then discarded by a shadowing `Vec::new()`
(transcribe.rs:195-201); only the token-overlap stitch (transcribe.rs:56-63) is sequential.
So rank r of R decodes the contiguous block [ceil(rK/R), ceil((r+1)K/R)) of the K windows,
every rank contributes one fixed-shape int32 buffer [ceil(K/R), 1 + row] (length + tokens)
to a single all-gather (RCCL over xGMI with backend "nccl"; gloo in CPU tests), and the
stitch is folded over all K rows in window order on the host -- identical to world size 1
by construction.  The reference has no distributed code: this is new.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np


def partition_windows(n_windows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of rank `rank`: [ceil(r K / R), ceil((r + 1) K / R))."""
    lo = -(-rank * n_windows // world)
    hi = -(-(rank + 1) * n_windows // world)
    return lo, min(hi, n_windows)


def rank_pcm_span(starts: Sequence[int], lens: Sequence[int], lo: int, hi: int) -> Tuple[int, int]:
    """Sample range [begin, end) of the waveform that windows [lo, hi) cover: the only PCM a rank needs resident
    (SURVEY.md section 8e).  Window i of the reference starts at i * shift (transcribe.rs:120-128), so inside the span the
    rank's windows are windows 0 .. hi - lo - 1 of `window_extents(end - begin)` with the same lengths -- the span may
    yield ONE more window than the rank owns (the 3 s overlap tail of its last window, which belongs to the next rank),
    hence callers pass win_end = hi - lo explicitly."""
    if hi <= lo:
        return 0, 0
    return int(starts[lo]), int(starts[hi - 1]) + int(lens[hi - 1])


def rows_per_rank(n_windows: int, world: int) -> int:
    return max(1, -(-n_windows // world))


def pack_rows(per_window: Sequence[Sequence[int]], n_rows: int, row_stride: int) -> np.ndarray:
    """[n_rows, 1 + row_stride] int32: column 0 = length (-1 marks an unused row)."""
    buf = np.zeros((n_rows, 1 + row_stride), dtype=np.int32)
    buf[:, 0] = -1
    for i, toks in enumerate(per_window):
        assert len(toks) <= row_stride
        buf[i, 0] = len(toks)
        buf[i, 1:1 + len(toks)] = toks
    return buf


def unpack_rows(gathered: np.ndarray, n_windows: int, world: int) -> List[List[int]]:
    """Inverse of pack_rows over the all-gathered [world, n_rows, 1 + row_stride] buffer, in window order."""
    out: List[List[int]] = []
    for r in range(world):
        lo, hi = partition_windows(n_windows, r, world)
        for i in range(hi - lo):
            n = int(gathered[r, i, 0])
            assert n >= 0, "rank %d row %d missing" % (r, i)
            out.append(gathered[r, i, 1:1 + n].tolist())
    assert len(out) == n_windows
    return out


def all_gather_rows(local: np.ndarray, world: int, device=None) -> np.ndarray:
    """One all-gather of the fixed-shape token buffer.  `device` selects the tensor device
    ("cuda:<i>" for RCCL, None/"cpu" for gloo)."""
    if world == 1:
        return local[None]
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(local)).reshape(-1)
    if device is not None and str(device) != "cpu":
        t = t.to(device)
    out = torch.empty(world * t.numel(), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)          # concatenation along dim 0 (gloo and nccl agree on this form)
    return out.cpu().numpy().reshape((world,) + tuple(local.shape))


def transcribe_sharded(decode_local: Callable[[int, int], List[List[int]]], stitch: Callable[[np.ndarray, np.ndarray], List[int]],
                       n_windows: int, rank: int, world: int, row_stride: int, device=None):
    """decode_local(lo, hi) -> per-window token lists of this rank's block; returns
    (stitched tokens over ALL windows, per-window tokens) on every rank."""
    lo, hi = partition_windows(n_windows, rank, world)
    local = decode_local(lo, hi) if hi > lo else []
    buf = pack_rows(local, rows_per_rank(n_windows, world), row_stride)
    per_window = unpack_rows(all_gather_rows(buf, world, device), n_windows, world)
    rows = np.zeros((n_windows, row_stride), dtype=np.int32)
    lens = np.zeros(n_windows, dtype=np.int32)
    for i, t in enumerate(per_window):
        rows[i, :len(t)] = t
        lens[i] = len(t)
    return stitch(rows, lens), per_window
