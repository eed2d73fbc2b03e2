"""Host-side mirror of the reference interface for the hot path, over the C ABI.

Same names, argument meaning and error behaviour as the reference seams
(/root/reference/src/audio.rs:34 prep_audio, src/model/mod.rs:47-71 Whisper::{forward,
forward_encoder, forward_decoder, encoder_ctx_size, decoder_ctx_size},
src/transcribe.rs:23-29 waveform_to_text), so the parity tests read like tests of the
reference would.  Shape-contract violations the reference `assert!`s on raise
`ShapeError` (a WbError with status WB_ERR_SHAPE).  All arithmetic happens in
libwhisper_hip.so; this file only marshals NumPy arrays.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import WbDecodeParams, WbDims, WbError, check
from .tokens import SpecialTokens

WB_F32, WB_BF16 = 0, 1
WB_ERR_SHAPE = -2


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _fp(a: np.ndarray):
    return a.ctypes.data_as(_lib.c_float_p)


def _ip(a: np.ndarray):
    return a.ctypes.data_as(_lib.c_int32_p)


def special_mask_bytes(whisper: "Whisper", is_special) -> np.ndarray:
    """uint8 [n_vocab] for the C side, which reads exactly n_vocab bytes.  The reference adds a [vocab_size] mask to
    [.., n_vocab] logits (transcribe.rs:243-275) and panics when the tokenizer's vocabulary and the model's differ; a
    shorter buffer here would be an out-of-bounds host read."""
    m = np.ascontiguousarray(is_special, dtype=np.uint8).reshape(-1)
    if m.shape[0] != whisper.dims["n_vocab"]:
        raise WbError(WB_ERR_SHAPE, f"special-token mask has {m.shape[0]} entries but the model's vocabulary has "
                                    f"{whisper.dims['n_vocab']} (tokenizer / checkpoint mismatch)")
    return m


def max_waveform_samples(n_frame_max: int) -> int:
    """audio.rs:12-17."""
    return int(_lib.load().wb_max_waveform_samples(int(n_frame_max)))


def prep_audio(waveform, sample_rate: float = 16000.0, device: int = 0) -> np.ndarray:
    """audio.rs:34: waveform [n_batch, n_samples] -> log-mel [n_batch, 80, n_samples // 160]."""
    lib = _lib.load()
    w = _f32(waveform)
    if w.ndim == 1:
        w = w[None]
    out = []
    for row in w:
        n = row.shape[0]
        mel = np.empty((80, max(n // 160, 0)), dtype=np.float32)
        nf = C.c_int64(0)
        check(lib.wb_prep_audio(device, _fp(row), n, float(sample_rate), _fp(mel), C.byref(nf)))
        out.append(mel)
    return np.stack(out)


def burn_record_tensors(mpk_gz_path: str, cfg_path: Optional[str] = None) -> Dict[str, np.ndarray]:
    """The tensors of a converted model file under their dump-directory names (host only, no GPU)."""
    out: Dict[str, np.ndarray] = {}

    def cb(_user, name, data, shape, rank):
        shp = tuple(int(shape[i]) for i in range(rank))
        n = int(np.prod(shp)) if shp else 1
        out[name.decode()] = np.ctypeslib.as_array(data, shape=(n,)).copy().reshape(shp)
        return 0

    fn = _lib.TENSOR_FN(cb)
    check(_lib.load().wb_burn_record_read(mpk_gz_path.encode(), cfg_path.encode() if cfg_path else None, fn, None))
    return out


def load_audio_waveform(path: str, any_rate: bool = False):
    """bin/transcribe/main.rs:31-55: (f32 samples, sample_rate); 16 kHz mono only, like the reference, unless
    `any_rate` (then the caller resamples: `resample`)."""
    lib = _lib.load()
    n, sr = C.c_int64(0), C.c_int32(0)
    check(lib.wb_wav_info(path.encode(), C.byref(n), C.byref(sr), None, None, None))
    out = np.empty(int(n.value), dtype=np.float32)
    got = C.c_int64(0)
    read = lib.wb_wav_read_f32_any_rate if any_rate else lib.wb_wav_read_f32
    check(read(path.encode(), _fp(out), int(n.value), C.byref(got)))
    return out[:int(got.value)], int(sr.value)


def resample_filter(rate_in: int, rate_out: int = 16000):
    """(taps f32 [20*max(up,down)+1], up, down) of the device resampler (host only)."""
    lib = _lib.load()
    n, up, down = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    check(lib.wb_resample_filter(rate_in, rate_out, None, 0, C.byref(n), C.byref(up), C.byref(down)))
    taps = np.empty(int(n.value), dtype=np.float32)
    check(lib.wb_resample_filter(rate_in, rate_out, _fp(taps), int(n.value), None, None, None))
    return taps, int(up.value), int(down.value)


def resample(pcm, rate_in: int, rate_out: int = 16000, device: int = 0) -> np.ndarray:
    """Polyphase resampling on the GPU (wb_resample_dev): f32 [n] -> f32 [ceil(n*rate_out/rate_in)]."""
    import torch
    lib = _lib.load()
    x = torch.as_tensor(np.ascontiguousarray(pcm, dtype=np.float32)).to(f"cuda:{device}")
    n_out = int(lib.wb_resample_len(x.numel(), rate_in, rate_out))
    if n_out < 0:
        raise ValueError(f"unsupported sample-rate pair {rate_in} -> {rate_out}")
    y = torch.empty(max(n_out, 1), dtype=torch.float32, device=x.device)
    torch.cuda.synchronize(x.device)
    got = C.c_int64(0)
    check(lib.wb_resample_dev(device, C.c_void_p(x.data_ptr()), x.numel(), rate_in, rate_out, C.c_void_p(y.data_ptr()),
                              y.numel(), C.byref(got)))
    return y[:int(got.value)].cpu().numpy()


def wav_info(path: str) -> dict:
    lib = _lib.load()
    n, sr, ch, bits, fl = C.c_int64(0), C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
    check(lib.wb_wav_info(path.encode(), C.byref(n), C.byref(sr), C.byref(ch), C.byref(bits), C.byref(fl)))
    return dict(n_samples=int(n.value), sample_rate=int(sr.value), channels=int(ch.value), bits=int(bits.value),
                is_float=bool(fl.value))


def pcm_s16_to_f32_dev(src_ptr: int, n: int, dst_ptr: int, device: int = 0) -> None:
    """main.rs:45-52 on the device: dst[i] = src[i] / 32767."""
    check(_lib.load().wb_pcm_s16_to_f32_dev(device, C.c_void_p(src_ptr), n, C.c_void_p(dst_ptr)))


def waveform_to_mels_dev(pcm_ptr: int, n_samples: int, starts, lens, mel_ptr: int, win_stride: int, row_stride: int,
                         sample_rate: float = 16000.0, clip_frames: int = 1490, padding: int = 10, device: int = 0,
                         iters: int = 1):
    """transcribe.rs:114-138 + :171-177 for a batch of windows, device pointers in and out.
    Returns (frames per window, HIP-event milliseconds of all `iters` passes)."""
    lib = _lib.load()
    st = np.ascontiguousarray(starts, dtype=np.int64)
    ln = np.ascontiguousarray(lens, dtype=np.int64)
    frames = np.zeros(len(st), dtype=np.int32)
    ms = C.c_double(0.0)
    check(lib.wb_waveform_to_mels_dev(device, C.c_void_p(pcm_ptr), n_samples, float(sample_rate),
                                      st.ctypes.data_as(_lib.c_int64_p), ln.ctypes.data_as(_lib.c_int64_p), len(st),
                                      clip_frames, padding, C.c_void_p(mel_ptr), win_stride, row_stride,
                                      frames.ctypes.data_as(_lib.c_int32_p), iters, C.byref(ms)))
    return frames, float(ms.value)


class Whisper:
    """mod.rs:41-71 `Whisper<B>` on one MI355X."""

    def __init__(self, handle, device: int):
        self._h = handle
        self.device = device
        d = WbDims()
        check(_lib.load().wb_model_dims(self._h, C.byref(d)))
        self.dims = {k: int(getattr(d, k)) for k, _ in WbDims._fields_}

    # -- construction -----------------------------------------------------------------
    @staticmethod
    def load_dump_dir(path: str, device: int = 0, compute_dtype: int = WB_F32) -> "Whisper":
        """load_whisper(path), load.rs:295-310."""
        h = C.c_void_p()
        check(_lib.load().wb_model_load_dump_dir(path.encode(), device, compute_dtype, C.byref(h)))
        return Whisper(h, device)

    @staticmethod
    def load_burn_record(mpk_gz_path: str, cfg_path: Optional[str] = None, device: int = 0,
                         compute_dtype: int = WB_F32) -> "Whisper":
        """load_whisper_model_file, bin/transcribe/main.rs:63-70 (+ the .cfg of :116-123)."""
        h = C.c_void_p()
        check(_lib.load().wb_model_load_burn_record(mpk_gz_path.encode(), cfg_path.encode() if cfg_path else None,
                                                    device, compute_dtype, C.byref(h)))
        return Whisper(h, device)

    @staticmethod
    def from_tensors(weights: Dict[str, np.ndarray], device: int = 0, compute_dtype: int = WB_F32) -> "Whisper":
        names = list(weights)
        arrs = [_f32(weights[k]) for k in names]
        n = len(names)
        c_names = (C.c_char_p * n)(*[k.encode() for k in names])
        c_data = (_lib.c_float_p * n)(*[_fp(a) for a in arrs])
        shp = [np.asarray(a.shape if a.ndim else (1,), dtype=np.int64) for a in arrs]
        c_shapes = (_lib.c_int64_p * n)(*[s.ctypes.data_as(_lib.c_int64_p) for s in shp])
        c_ranks = (C.c_int32 * n)(*[len(s) for s in shp])
        h = C.c_void_p()
        check(_lib.load().wb_model_load_tensors(c_names, c_data, c_shapes, c_ranks, n, device, compute_dtype,
                                                C.byref(h)))
        return Whisper(h, device)

    def close(self):
        if self._h:
            _lib.load().wb_model_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_layernorm_variant(self, eps_inside_sqrt: bool):
        check(_lib.load().wb_model_set_ln_variant(self._h, int(eps_inside_sqrt)))

    def set_frame_limit(self, whisper_geometry: bool):
        """False (default): a window holds at most n_audio_ctx MEL FRAMES, as the reference asserts (mod.rs:236-241).
        True: at most n_audio_ctx encoder positions = 2 n_audio_ctx frames -- Whisper's own 30 s window (the "perf
        geometry" T = 3000, C = 1500 of SURVEY 8d config 2b); the reference panics on such a window."""
        check(_lib.load().wb_model_set_frame_limit(self._h, int(bool(whisper_geometry))))
        self._frame_limit_x2 = bool(whisper_geometry)

    def encoder_gemm(self) -> str:
        """Arithmetic of the encoder-side Linear layers: "f32" (exact-f32 MFMA), "f16x3" (split precision: three fp16 MFMAs
        per product, f32-grade; the default)."""
        return ("f32", "f16x3")[_lib.load().wb_model_encoder_gemm(self._h)]

    def decoder_gemm(self) -> str:
        """Arithmetic of the decoder's Linear layers in batch mode: "f16x3" (split precision on fp16 weight tiles; default)
        or "f32" (exact-f32 MFMA; also after a decoder range guard has tripped -- wb_model_decoder_gemm)."""
        return ("f32", "f16x3")[_lib.load().wb_model_decoder_gemm(self._h)]

    def max_mel_frames(self) -> int:
        """Mel frames one window may hold (what transcribe.rs:32 calls n_ctx_max_encoder)."""
        return self.dims["n_audio_ctx"] * (2 if getattr(self, "_frame_limit_x2", False) else 1)

    # -- mod.rs:47-71 -------------------------------------------------------------------
    def encoder_ctx_size(self) -> int:
        return self.dims["n_audio_ctx"]

    def decoder_ctx_size(self) -> int:
        return self.dims["n_text_ctx"]

    def forward_encoder(self, mel) -> np.ndarray:
        """[B, 80, T] -> [B, C, d]."""
        mel = _f32(mel)
        B, n_mels, T = mel.shape
        if n_mels != self.dims["n_mels"]:     # mod.rs:231-235
            raise WbError(WB_ERR_SHAPE, f"Audio mel spectrum size must be {self.dims['n_mels']}.")
        out = np.empty((B, (T - 1) // 2 + 1 if T > 0 else 0, self.dims["n_audio_state"]), dtype=np.float32)
        check(_lib.load().wb_forward_encoder(self._h, _fp(mel), B, T, _fp(out)))
        return out

    def forward_decoder(self, tokens, encoder_output) -> np.ndarray:
        """tokens [n, L] int, encoder_output [n, C, d] -> logits [n, L, V]."""
        tokens = _i32(tokens)
        enc = _f32(encoder_output)
        n, L = tokens.shape
        assert enc.shape[0] == n and enc.shape[2] == self.dims["n_text_state"]
        logits = np.empty((n, L, self.dims["n_vocab"]), dtype=np.float32)
        check(_lib.load().wb_forward_decoder(self._h, _ip(tokens), n, L, _fp(enc), enc.shape[1], _fp(logits)))
        return logits

    def forward(self, mel, tokens) -> np.ndarray:
        mel = _f32(mel)
        tokens = _i32(tokens)
        B, _, T = mel.shape
        logits = np.empty((B, tokens.shape[1], self.dims["n_vocab"]), dtype=np.float32)
        check(_lib.load().wb_forward(self._h, _fp(mel), B, T, _ip(tokens), tokens.shape[1], _fp(logits)))
        return logits


def decode_params(st: SpecialTokens, beam_size: int = 5, max_depth: int = 100, **kw) -> WbDecodeParams:
    """The constants transcribe.rs hard-codes (beam 5 x depth 100, padding 10, 3 s overlap, (40, 3) stitch)."""
    p = WbDecodeParams()
    _lib.load().wb_decode_params_default(C.byref(p))
    p.beam_size, p.max_depth = beam_size, max_depth
    p.tok_start_of_transcript, p.tok_language = st.start_of_transcript, st.language
    p.tok_transcribe, p.tok_no_timestamps, p.tok_end_of_text = st.transcribe, st.no_timestamps, st.end_of_text
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def window_extents(n_samples: int, sample_rate: int, window_len: int, overlap_seconds: int = 3):
    """transcribe.rs:120-128."""
    lib = _lib.load()
    n = int(lib.wb_window_extents(n_samples, sample_rate, window_len, overlap_seconds, None, None, 0))
    starts = np.zeros(n, dtype=np.int64)
    lens = np.zeros(n, dtype=np.int64)
    lib.wb_window_extents(n_samples, sample_rate, window_len, overlap_seconds,
                          starts.ctypes.data_as(_lib.c_int64_p), lens.ctypes.data_as(_lib.c_int64_p), n)
    return starts, lens


def find_chunk_overlap(prev_tokens: Sequence[int], curr_tokens: Sequence[int], max_n_offsets: int,
                       min_n_overlaps: int) -> Optional[Tuple[int, int]]:
    """transcribe.rs:76-110."""
    p, c = _i32(list(prev_tokens)), _i32(list(curr_tokens))
    pi, ci = C.c_int64(0), C.c_int64(0)
    r = _lib.load().wb_find_chunk_overlap(_ip(p), len(p), _ip(c), len(c), max_n_offsets, min_n_overlaps,
                                          C.byref(pi), C.byref(ci))
    return (int(pi.value), int(ci.value)) if r == 1 else None


def stitch_windows(win_tokens: np.ndarray, win_lens: np.ndarray, max_n_offsets: int = 40,
                   min_n_overlaps: int = 3) -> List[int]:
    """Fold transcribe.rs:56-63 over per-window token rows in window order."""
    wt, wl = _i32(win_tokens), _i32(win_lens)
    cap = int(wl.sum()) + 1
    out = np.zeros(cap, dtype=np.int32)
    n_out = C.c_int64(0)
    check(_lib.load().wb_stitch_windows(_ip(wt), wt.shape[1] if wt.ndim == 2 else 0, _ip(wl), len(wl),
                                        max_n_offsets, min_n_overlaps, _ip(out), cap, C.byref(n_out)))
    return out[:n_out.value].tolist()


def waveform_to_tokens(whisper: Whisper, st: SpecialTokens, waveform, sample_rate: int = 16000,
                       beam_size: int = 5, max_depth: int = 100, win_begin: int = 0, win_end: int = -1,
                       params: Optional[WbDecodeParams] = None, device_ptr: Optional[int] = None,
                       n_samples: Optional[int] = None):
    """waveform_to_text (transcribe.rs:23-74) without the tokenizer.

    Returns (stitched token ids of the local windows, per-window token lists).  [win_begin, win_end)
    selects the windows this process decodes (multi-GPU sharding); default all.
    With `device_ptr` (+ `n_samples`) the waveform is read in place from device memory
    (wb_waveform_to_tokens_dev) and `waveform` is ignored."""
    lib = _lib.load()
    if device_ptr is None:
        wav = _f32(waveform).reshape(-1)
        n_samples = len(wav)
    p = params or decode_params(st, beam_size, max_depth)
    wlen = max_waveform_samples(whisper.max_mel_frames() - p.padding)
    starts, _ = window_extents(n_samples, sample_rate, wlen, p.overlap_seconds)
    n_win = len(starts)
    if win_end < 0:
        win_end = n_win
    n_local = max(0, win_end - win_begin)
    stride = 4 + p.max_depth + 4
    win_tokens = np.zeros((max(n_local, 1), stride), dtype=np.int32)
    win_lens = np.zeros(max(n_local, 1), dtype=np.int32)
    cap = max(n_local, 1) * stride
    stitched = np.zeros(cap, dtype=np.int32)
    n_st = C.c_int64(0)
    mask = special_mask_bytes(whisper, st.is_special)
    if device_ptr is None:
        check(lib.wb_waveform_to_tokens(whisper._h, _fp(wav), n_samples, sample_rate, C.byref(p),
                                        mask.ctypes.data_as(_lib.c_uint8_p), win_begin, win_end, _ip(win_tokens),
                                        stride, _ip(win_lens), _ip(stitched), cap, C.byref(n_st)))
    else:
        check(lib.wb_waveform_to_tokens_dev(whisper._h, C.c_void_p(device_ptr), n_samples, sample_rate, C.byref(p),
                                            mask.ctypes.data_as(_lib.c_uint8_p), win_begin, win_end,
                                            _ip(win_tokens), stride, _ip(win_lens), _ip(stitched), cap,
                                            C.byref(n_st)))
    per_window = [win_tokens[i, :win_lens[i]].tolist() for i in range(n_local)]
    return stitched[:n_st.value].tolist(), per_window


def waveform_to_text(whisper: Whisper, bpe, lang, waveform, sample_rate: int = 16000):
    """transcribe.rs:23-29 shape: returns (text, tokens).  `bpe` must offer
    `special_tokens(lang) -> SpecialTokens` and `decode(tokens, skip_special) -> str`
    (the tokenizer stays outside the engine; none ships in this environment)."""
    st = bpe.special_tokens(lang)
    tokens, _ = waveform_to_tokens(whisper, st, waveform, sample_rate)
    return bpe.decode(tokens, True), tokens


class Session:
    """KV-cached decode session over a batch of windows (include/whisper_hip.h, wb_session_*).

    New relative to the reference, which has no KV cache (transcribe.rs:270); result-equivalent
    to running `beamsearch_next` (transcribe.rs:253-307) on the same beams."""

    def __init__(self, whisper: Whisper, handle, n_windows: int):
        self._w = whisper
        self._h = handle
        self.n_windows = n_windows

    @staticmethod
    def begin(whisper: Whisper, waveform, starts, lens, max_beams: int = 5, padding: int = 10) -> "Session":
        wav = _f32(waveform).reshape(-1)
        st = np.ascontiguousarray(starts, dtype=np.int64)
        ln = np.ascontiguousarray(lens, dtype=np.int64)
        h = C.c_void_p()
        check(_lib.load().wb_session_begin(whisper._h, _fp(wav), len(wav), st.ctypes.data_as(_lib.c_int64_p),
                                           ln.ctypes.data_as(_lib.c_int64_p), len(st), max_beams, padding,
                                           C.byref(h)))
        return Session(whisper, h, len(st))

    @staticmethod
    def begin_mel(whisper: Whisper, mels: Sequence[np.ndarray], max_beams: int = 5, padding: int = 10) -> "Session":
        mels = [_f32(m) for m in mels]
        T = _i32([m.shape[1] for m in mels])
        flat = np.concatenate([m.reshape(-1) for m in mels])
        h = C.c_void_p()
        check(_lib.load().wb_session_begin_mel(whisper._h, _fp(flat), _ip(T), len(mels), max_beams, padding,
                                               C.byref(h)))
        return Session(whisper, h, len(mels))

    def set_special_mask(self, is_special) -> None:
        m = special_mask_bytes(self._w, is_special)
        check(_lib.load().wb_session_set_special_mask(self._h, m.ctypes.data_as(_lib.c_uint8_p)))

    def step(self, new_tokens, parent, window, apply_special_mask: bool = False, k: int = 5):
        t, p, w = _i32(new_tokens), _i32(parent), _i32(window)
        n = len(t)
        ids = np.zeros((n, max(k, 1)), dtype=np.int32)
        lps = np.zeros((n, max(k, 1)), dtype=np.float32)
        check(_lib.load().wb_session_step(self._h, _ip(t), _ip(p), _ip(w), n, int(apply_special_mask), k,
                                          _ip(ids), _fp(lps)))
        return (ids[:, :k], lps[:, :k]) if k > 0 else (None, None)

    def last_logprobs(self, slot: int) -> np.ndarray:
        out = np.empty(self._w.dims["n_vocab"], dtype=np.float32)
        check(_lib.load().wb_session_last_logprobs(self._h, slot, _fp(out)))
        return out

    def encoder_output(self, w: int) -> np.ndarray:
        c = C.c_int32(0)
        check(_lib.load().wb_session_encoder_output(self._h, w, None, C.byref(c)))
        out = np.empty((c.value, self._w.dims["n_audio_state"]), dtype=np.float32)
        check(_lib.load().wb_session_encoder_output(self._h, w, _fp(out), C.byref(c)))
        return out

    def decode(self, params: WbDecodeParams) -> List[List[int]]:
        stride = 4 + params.max_depth + 4
        toks = np.zeros((self.n_windows, stride), dtype=np.int32)
        lens = np.zeros(self.n_windows, dtype=np.int32)
        check(_lib.load().wb_session_decode(self._h, C.byref(params), _ip(toks), stride, _ip(lens)))
        return [toks[i, :lens[i]].tolist() for i in range(self.n_windows)]

    def close(self):
        if self._h:
            _lib.load().wb_session_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
