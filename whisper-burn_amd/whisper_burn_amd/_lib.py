"""ctypes binding of libwhisper_hip.so (include/whisper_hip.h).

There is no CPU fallback: if the HIP library is missing or fails to load, importing
the engine raises -- the product path never routes through the oracle.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libwhisper_hip.so")

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)
c_uint8_p = C.POINTER(C.c_uint8)
c_double_p = C.POINTER(C.c_double)


class WbDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
        "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")]


class WbDecodeParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "beam_size", "max_depth", "padding", "overlap_seconds", "max_n_offsets", "min_n_overlaps",
        "mask_until_len", "max_batch_windows",
        "tok_start_of_transcript", "tok_language", "tok_transcribe", "tok_no_timestamps",
        "tok_end_of_text")]


# wb_step_fn (include/whisper_hip.h)
STEP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, c_int32_p, c_int32_p, c_int32_p, C.c_int, C.c_int, C.c_int,
                      c_int32_p, c_float_p)

# wb_allgather_fn (include/whisper_hip.h)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)

# every symbol include/whisper_hip.h declares: name -> (restype, argtypes)
TENSOR_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, c_float_p, c_int64_p, C.c_int32)

SIGNATURES = {
    "wb_model_load_dump_dir": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "wb_model_load_tensors": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(c_float_p), C.POINTER(c_int64_p),
                                        c_int32_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "wb_model_dims": (C.c_int, [C.c_void_p, C.POINTER(WbDims)]),
    "wb_model_free": (None, [C.c_void_p]),
    "wb_model_set_ln_variant": (C.c_int, [C.c_void_p, C.c_int]),
    "wb_model_set_frame_limit": (C.c_int, [C.c_void_p, C.c_int]),
    "wb_model_encoder_gemm": (C.c_int, [C.c_void_p]),
    "wb_model_decoder_gemm": (C.c_int, [C.c_void_p]),
    "wb_max_waveform_samples": (C.c_int64, [C.c_int64]),
    "wb_prep_audio": (C.c_int, [C.c_int, c_float_p, C.c_int64, C.c_double, c_float_p, c_int64_p]),
    "wb_model_load_burn_record": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "wb_burn_record_read": (C.c_int, [C.c_char_p, C.c_char_p, TENSOR_FN, C.c_void_p]),
    "wb_wav_info": (C.c_int, [C.c_char_p, c_int64_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p]),
    "wb_wav_read_f32": (C.c_int, [C.c_char_p, c_float_p, C.c_int64, c_int64_p]),
    "wb_wav_read_f32_any_rate": (C.c_int, [C.c_char_p, c_float_p, C.c_int64, c_int64_p]),
    "wb_resample_len": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32]),
    "wb_resample_filter": (C.c_int, [C.c_int32, C.c_int32, c_float_p, C.c_int32, c_int32_p, c_int32_p, c_int32_p]),
    "wb_resample_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, c_int64_p]),
    "wb_pcm_s16_to_f32_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "wb_waveform_to_mels_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_double, c_int64_p, c_int64_p, C.c_int32,
                                          C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, c_int32_p, C.c_int32,
                                          c_double_p]),
    "wb_forward_encoder": (C.c_int, [C.c_void_p, c_float_p, C.c_int, C.c_int, c_float_p]),
    "wb_forward_decoder": (C.c_int, [C.c_void_p, c_int32_p, C.c_int, C.c_int, c_float_p, C.c_int, c_float_p]),
    "wb_forward": (C.c_int, [C.c_void_p, c_float_p, C.c_int, C.c_int, c_int32_p, C.c_int, c_float_p]),
    "wb_decode_params_default": (None, [C.POINTER(WbDecodeParams)]),
    "wb_comm_unique_id": (C.c_int, [c_uint8_p]),
    "wb_comm_init": (C.c_int, [c_uint8_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "wb_comm_free": (None, [C.c_void_p]),
    "wb_comm_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "wb_shard_partition": (C.c_int, [C.c_int64, C.c_int, C.c_int, c_int64_p, c_int64_p]),
    "wb_waveform_to_tokens_sharded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.POINTER(WbDecodeParams),
                                               c_uint8_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, c_int32_p, C.c_int32,
                                               c_int32_p, C.c_int64, c_int32_p, C.c_int64, c_int64_p]),
    "wb_session_begin": (C.c_int, [C.c_void_p, c_float_p, C.c_int64, c_int64_p, c_int64_p, C.c_int, C.c_int,
                                   C.c_int, C.POINTER(C.c_void_p)]),
    "wb_session_begin_mel": (C.c_int, [C.c_void_p, c_float_p, c_int32_p, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(C.c_void_p)]),
    "wb_session_set_special_mask": (C.c_int, [C.c_void_p, c_uint8_p]),
    "wb_session_step": (C.c_int, [C.c_void_p, c_int32_p, c_int32_p, c_int32_p, C.c_int, C.c_int, C.c_int,
                                  c_int32_p, c_float_p]),
    "wb_session_last_logprobs": (C.c_int, [C.c_void_p, C.c_int, c_float_p]),
    "wb_session_encoder_output": (C.c_int, [C.c_void_p, C.c_int, c_float_p, c_int32_p]),
    "wb_session_free": (None, [C.c_void_p]),
    "wb_session_decode": (C.c_int, [C.c_void_p, C.POINTER(WbDecodeParams), c_int32_p, C.c_int32, c_int32_p]),
    "wb_session_decode_prompt": (C.c_int, [C.c_void_p, C.POINTER(WbDecodeParams), c_int32_p, C.c_int32, c_int32_p, C.c_int32,
                                           c_int32_p]),
    "wb_waveform_to_tokens_prompted": (C.c_int, [C.c_void_p, c_float_p, C.c_int64, C.c_int, C.POINTER(WbDecodeParams),
                                                 c_uint8_p, C.c_int32, C.c_int32, c_int32_p, C.c_int32, c_int32_p,
                                                 c_int32_p, C.c_int64, c_int64_p]),
    "wb_beam_search_device": (C.c_int, [C.c_int, C.POINTER(WbDecodeParams), C.c_int, C.c_int, C.c_void_p, C.c_void_p, c_int32_p,
                                        C.c_int32, c_int32_p]),
    "wb_beam_search": (C.c_int, [C.POINTER(WbDecodeParams), C.c_int, C.c_int, C.c_void_p, C.c_void_p, c_int32_p,
                                 C.c_int32, c_int32_p]),
    "wb_waveform_to_tokens": (C.c_int, [C.c_void_p, c_float_p, C.c_int64, C.c_int, C.POINTER(WbDecodeParams),
                                        c_uint8_p, C.c_int, C.c_int, c_int32_p, C.c_int32, c_int32_p,
                                        c_int32_p, C.c_int64, c_int64_p]),
    "wb_waveform_to_tokens_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(WbDecodeParams),
                                            c_uint8_p, C.c_int, C.c_int, c_int32_p, C.c_int32, c_int32_p,
                                            c_int32_p, C.c_int64, c_int64_p]),
    "wb_window_extents": (C.c_int64, [C.c_int64, C.c_int, C.c_int64, C.c_int, c_int64_p, c_int64_p, C.c_int64]),
    "wb_find_chunk_overlap": (C.c_int, [c_int32_p, C.c_int64, c_int32_p, C.c_int64, C.c_int, C.c_int,
                                        c_int64_p, c_int64_p]),
    "wb_stitch_windows": (C.c_int, [c_int32_p, C.c_int32, c_int32_p, C.c_int, C.c_int, C.c_int, c_int32_p,
                                    C.c_int64, c_int64_p]),
    "wb_first_repetition_end": (C.c_int64, [c_int32_p, C.c_int64, C.c_int64]),
    "wb_repetition_period": (C.c_int64, [c_int32_p, C.c_int64, C.c_int64]),
    "wb_find_repeated_tokens_index": (C.c_int, [c_int32_p, C.c_int64, C.c_int64, C.c_int64, c_int64_p, c_int64_p]),
    "wb_mel_constants": (C.c_int, [C.c_double, c_float_p, c_float_p]),
    "wb_profile_enable": (C.c_int, [C.c_int]),
    "wb_profile_read": (C.c_int, [c_double_p, C.c_int]),
    "wb_profile_kernels": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "wb_last_error": (C.c_char_p, []),
    "wb_version": (C.c_char_p, []),
}



class WbKernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("calls", C.c_int64), ("total_ms", C.c_double), ("algo_bytes", C.c_double)]


def profile_kernels(reset: bool = True):
    """wb_profile_kernels as a list of dicts (name, calls, total_ms, algo_bytes)."""
    lib = load()
    buf = (WbKernelStat * 32)()
    n = min(32, int(lib.wb_profile_kernels(C.cast(buf, C.c_void_p), 32, int(reset))))
    return [dict(name=buf[i].name.decode(), calls=int(buf[i].calls), total_ms=float(buf[i].total_ms),
                 algo_bytes=float(buf[i].algo_bytes)) for i in range(n)]


_lib = None


def load(path: str | None = None) -> C.CDLL:
    """Load libwhisper_hip.so and bind every declared symbol (raises if anything is missing)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("WHISPER_HIP_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise OSError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      f"(make -C whisper-burn_amd/csrc); there is no CPU fallback")
    try:  # if PyTorch-ROCm is in the process, let it bring ITS libamdhip64.so.7 first (one HIP runtime)
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional plumbing
        pass
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if b"hipemu" in lib.wb_version() and os.environ.get("WHISPER_HIP_ALLOW_EMU") != "1":
        # tools/hipemu builds the same sources against a functional model of the GPU for kernel development on
        # GPU-less machines; it is test infrastructure and never a product path
        raise OSError(f"{path} is the hipemu functional-model build, not the gfx950 library; the engine has no CPU "
                      f"path (tests that drive the emulator set WHISPER_HIP_ALLOW_EMU=1)")
    _lib = lib
    return lib


class WbError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"whisper_hip status {status}: {msg}")
        self.status = status


def check(status: int) -> None:
    if status != 0:
        raise WbError(status, (load().wb_last_error() or b"").decode("utf-8", "replace"))
