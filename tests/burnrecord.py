"""The reference's converted model file (python side, fixtures only).

/root/reference/src/bin/convert/main.rs:17-19, :51 writes `<name>.mpk.gz` with Burn's
NamedMpkGzFileRecorder<FullPrecisionSettings>: gzip(MessagePack, structs as maps with field names) of
{metadata, item = the Whisper module record}, the record mirroring the module tree of src/model/mod.rs;
`<name>.cfg` is the WhisperConfig as JSON (mod.rs:16-20).  Burn 0.9.0 is not vendored with the reference,
so the nesting under a Param cannot be checked offline; `write_burn_record` can emit both plausible
nestings and the native reader (csrc/record_load.cpp) is structural.  Format parity: unpinned.
"""
from __future__ import annotations

import gzip
import json
import uuid

import msgpack
import numpy as np


def _param(arr: np.ndarray, nesting: str) -> dict:
    arr = np.asarray(arr, dtype=np.float32)
    data = {"value": arr.reshape(-1).tolist(), "shape": [int(s) for s in arr.shape]}
    if nesting == "data":
        data = {"data": data}
    return {"id": str(uuid.uuid4()), "param": data}


def _linear(w: dict, base: str, nesting: str) -> dict:
    return {"weight": _param(w[base + "/weight"], nesting),
            "bias": _param(w[base + "/bias"], nesting) if base + "/bias" in w else None}


def _ln(w: dict, base: str, nesting: str) -> dict:
    return {"gamma": _param(w[base + "/weight"], nesting), "beta": _param(w[base + "/bias"], nesting),
            "epsilon": float(np.asarray(w[base + "/eps"]).reshape(-1)[0])}


def _attn(w: dict, base: str, nesting: str) -> dict:
    return {"n_head": int(np.asarray(w[base + "/n_head"]).reshape(-1)[0]),
            **{k: _linear(w, f"{base}/{k}", nesting) for k in ("query", "key", "value", "out")}}


def _mlp(w: dict, base: str, nesting: str) -> dict:
    return {"lin1": _linear(w, base + "/mlp1", nesting), "gelu": None, "lin2": _linear(w, base + "/mlp2", nesting)}


def module_record(w: dict, nesting: str = "flat") -> dict:
    """The Whisper module record (mod.rs:42-45 and below) from dump-style tensors."""
    n_enc = int(np.asarray(w["encoder/n_layer"]).reshape(-1)[0])
    n_dec = int(np.asarray(w["decoder/n_layer"]).reshape(-1)[0])
    d = int(np.asarray(w["encoder/n_audio_state"]).reshape(-1)[0])
    n_ctx = w["decoder/positional_embedding"].shape[0]
    enc = {
        "conv1": _linear(w, "encoder/conv1", nesting), "gelu1": None,
        "conv2": _linear(w, "encoder/conv2", nesting), "gelu2": None,
        "blocks": [{"attn": _attn(w, f"encoder/block_{i}/attn", nesting),
                    "attn_ln": _ln(w, f"encoder/block_{i}/attn_ln", nesting),
                    "mlp": _mlp(w, f"encoder/block_{i}/mlp", nesting),
                    "mlp_ln": _ln(w, f"encoder/block_{i}/mlp_ln", nesting)} for i in range(n_enc)],
        "ln_post": _ln(w, "encoder/ln_post", nesting),
        "positional_embedding": _param(w["encoder/positional_embedding"], nesting),
        "n_mels": 80, "n_audio_ctx": int(w["encoder/positional_embedding"].shape[0]),
    }
    mask = np.triu(np.full((n_ctx, n_ctx), -np.inf, dtype=np.float32), k=1)          # mod.rs:535-544
    mask = np.where(np.isinf(mask), np.float32(-1e30), mask)                        # (msgpack floats: keep it finite)
    dec = {
        "token_embedding": _param(w["decoder/token_embedding/weight"], nesting),
        "positional_embedding": _param(w["decoder/positional_embedding"], nesting),
        "blocks": [{"attn": _attn(w, f"decoder/block_{i}/attn", nesting),
                    "attn_ln": _ln(w, f"decoder/block_{i}/attn_ln", nesting),
                    "cross_attn": _attn(w, f"decoder/block_{i}/cross_attn", nesting),
                    "cross_attn_ln": _ln(w, f"decoder/block_{i}/cross_attn_ln", nesting),
                    "mlp": _mlp(w, f"decoder/block_{i}/mlp", nesting),
                    "mlp_ln": _ln(w, f"decoder/block_{i}/mlp_ln", nesting)} for i in range(n_dec)],
        "ln": _ln(w, "decoder/ln", nesting),
        "mask": _param(mask, nesting),
        "n_vocab": int(w["decoder/token_embedding/weight"].shape[0]), "n_text_ctx": int(n_ctx),
    }
    assert enc["conv1"]["weight"]["param"] is not None and d > 0
    return {"encoder": enc, "decoder": dec}


def write_burn_record(w: dict, mpk_gz_path: str, cfg_path: str | None = None, nesting: str = "flat") -> None:
    rec = {"metadata": {"float": "f32", "int": "i32",
                        "format": "burn_core::record::file::NamedMpkGzFileRecorder<burn_core::record::settings::FullPrecisionSettings>",
                        "version": "0.9.0", "settings": "FullPrecisionSettings"},
           "item": module_record(w, nesting)}
    with gzip.open(mpk_gz_path, "wb", compresslevel=1) as f:
        f.write(msgpack.packb(rec, use_single_float=True))
    if cfg_path:
        d = int(np.asarray(w["encoder/n_audio_state"]).reshape(-1)[0])
        cfg = {"audio_encoder_config": {"n_mels": 80, "n_audio_ctx": int(w["encoder/positional_embedding"].shape[0]),
                                        "n_audio_state": d,
                                        "n_audio_head": int(np.asarray(w["encoder/block_0/attn/n_head"]).reshape(-1)[0]),
                                        "n_audio_layer": int(np.asarray(w["encoder/n_layer"]).reshape(-1)[0])},
               "text_decoder_config": {"n_vocab": int(w["decoder/token_embedding/weight"].shape[0]),
                                       "n_text_ctx": int(w["decoder/positional_embedding"].shape[0]), "n_text_state": d,
                                       "n_text_head": int(np.asarray(w["decoder/block_0/attn/n_head"]).reshape(-1)[0]),
                                       "n_text_layer": int(np.asarray(w["decoder/n_layer"]).reshape(-1)[0])}}
        with open(cfg_path, "w") as f:
            json.dump(cfg, f)
