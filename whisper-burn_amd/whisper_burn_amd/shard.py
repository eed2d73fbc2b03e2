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

from typing import Callable, List, Optional, Sequence, Tuple

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


# ---- the same path behind the C ABI (csrc/shard.cpp): what a Rust / C caller of libwhisper_hip.so uses ---------------------
class RcclComm:
    """wb_comm: the built-in RCCL transport of wb_waveform_to_tokens_sharded.  Rank 0 calls `RcclComm.unique_id()` and ships
    the 128 bytes to the other ranks by any means; every rank then constructs RcclComm(id, rank, world, device)."""

    def __init__(self, unique_id: bytes, rank: int, world: int, device: int = 0):
        import ctypes as C
        from . import _lib
        self._h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        _lib.check(_lib.load().wb_comm_init(C.cast(buf, _lib.c_uint8_p), rank, world, device, C.byref(self._h)))
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import _lib
        buf = (C.c_uint8 * 128)()
        _lib.check(_lib.load().wb_comm_unique_id(C.cast(buf, _lib.c_uint8_p)))
        return bytes(buf)

    def allgather(self, local: np.ndarray) -> np.ndarray:
        """[world, *local.shape] -- wb_comm_allgather on a host array (int32 / float32 payloads: 4-byte units)."""
        import ctypes as C
        from . import _lib
        local = np.ascontiguousarray(local)
        out = np.empty((self.world,) + local.shape, dtype=local.dtype)
        _lib.check(_lib.load().wb_comm_allgather(self._h, local.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                                 local.nbytes))
        return out

    def close(self):
        if self._h:
            from . import _lib
            _lib.load().wb_comm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def c_partition_windows(n_windows: int, rank: int, world: int) -> Tuple[int, int]:
    """wb_shard_partition (the C ABI's statement of partition_windows)."""
    import ctypes as C
    from . import _lib
    lo, hi = C.c_int64(0), C.c_int64(0)
    _lib.check(_lib.load().wb_shard_partition(n_windows, rank, world, C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


def waveform_to_tokens_sharded(whisper, st, waveform, rank: int, world: int, comm=None, allgather=None,
                               sample_rate: int = 16000, params=None, device_ptr: Optional[int] = None,
                               n_samples: Optional[int] = None):
    """wb_waveform_to_tokens_sharded: this rank decodes its block of windows, ONE all-gather, stitch on every rank.

    The exchange is `comm` (an RcclComm: RCCL over xGMI, inside the library) or `allgather` (a Python callable
    `f(local: np.ndarray[int32]) -> np.ndarray[world, ...]`, e.g. over torch.distributed / gloo); neither is needed at
    world == 1.  Returns (stitched tokens over ALL windows, per-window token lists of ALL windows)."""
    import ctypes as C
    from . import _lib
    from .model import _f32, _ip, decode_params, max_waveform_samples, special_mask_bytes, window_extents
    lib = _lib.load()
    p = params or decode_params(st)
    if device_ptr is None:
        wav = _f32(waveform).reshape(-1)
        n_samples = len(wav)
        pcm = wav.ctypes.data_as(C.c_void_p)
    else:
        pcm = C.c_void_p(device_ptr)
    wlen = max_waveform_samples(whisper.max_mel_frames() - p.padding)
    K = len(window_extents(n_samples, sample_rate, wlen, p.overlap_seconds)[0])
    stride = 4 + p.max_depth + 4
    win_tokens = np.zeros((max(K, 1), stride), dtype=np.int32)
    win_lens = np.zeros(max(K, 1), dtype=np.int32)
    cap = max(K, 1) * stride
    stitched = np.zeros(cap, dtype=np.int32)
    n_st = C.c_int64(0)
    mask = special_mask_bytes(whisper, st.is_special)
    fn, user, keep = None, None, None
    if comm is not None:
        fn, user = C.cast(lib.wb_comm_allgather, C.c_void_p), comm._h
    elif allgather is not None:
        def thunk(_user, send, recv, nbytes):
            try:
                local = np.ctypeslib.as_array(C.cast(send, _lib.c_int32_p), shape=(nbytes // 4,)).copy()
                out = np.ascontiguousarray(allgather(local), dtype=np.int32).reshape(-1)
                assert out.size * 4 == nbytes * world, (out.size, nbytes, world)
                C.memmove(recv, out.ctypes.data, out.nbytes)
                return 0
            except Exception:                       # (an exception must not unwind through the C frames)
                import traceback
                traceback.print_exc()
                return -4
        keep = _lib.ALLGATHER_FN(thunk)
        fn = C.cast(keep, C.c_void_p)
    _lib.check(lib.wb_waveform_to_tokens_sharded(whisper._h, pcm, 0 if device_ptr is None else 1, n_samples, sample_rate,
                                                 C.byref(p), mask.ctypes.data_as(_lib.c_uint8_p), rank, world, fn, user,
                                                 _ip(win_tokens), stride, _ip(win_lens), max(K, 1), _ip(stitched), cap,
                                                 C.byref(n_st)))
    del keep
    return stitched[:n_st.value].tolist(), [win_tokens[i, :win_lens[i]].tolist() for i in range(K)]
