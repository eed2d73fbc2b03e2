"""Seeded synthetic Whisper weights and 16 kHz audio at the real shapes.

No Whisper checkpoint, tokenizer.json or network exists in the build/run
environment, so parity and throughput are established on seeded synthetic
weights written in the reference's dump-directory tensor naming
(/root/reference/src/model/load.rs:203-310, python/dump.py:130-210) and seeded
synthetic audio. NumPy/SciPy only.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
from scipy.signal import lfilter

# name -> (n_state, n_head, n_layer, n_vocab); n_mels=80, n_audio_ctx=1500, n_text_ctx=448
PRESETS = OrderedDict([
    ("tiny.en", (384, 6, 4, 51864)),
    ("base.en", (512, 8, 6, 51864)),
    ("small", (768, 12, 12, 51865)),
    ("medium", (1024, 16, 24, 51865)),
    ("large-v2", (1280, 20, 32, 51865)),
])
_ALIASES = {"tiny_en": "tiny.en", "base_en": "base.en", "large_v2": "large-v2", "large": "large-v2"}


def preset_dims(name: str) -> dict:
    name = _ALIASES.get(name, name)
    d, h, n_layer, v = PRESETS[name]
    return dict(n_mels=80, n_audio_ctx=1500, n_audio_state=d, n_audio_head=h, n_audio_layer=n_layer,
                n_vocab=v, n_text_ctx=448, n_text_state=d, n_text_head=h, n_text_layer=n_layer)


def preset_seed(name: str) -> int:
    name = _ALIASES.get(name, name)
    return 0x5EED0000 + list(PRESETS).index(name)


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> np.ndarray:
    """The fixed sinusoid table real Whisper checkpoints carry as encoder.positional_embedding."""
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = np.exp(-inc * np.arange(channels // 2, dtype=np.float64))
    t = np.arange(length, dtype=np.float64)[:, None] * inv[None, :]
    return np.concatenate([np.sin(t), np.cos(t)], axis=1).astype(np.float32)


def synth_weights(dims: dict, seed: int, logit_scale: float = 6.0) -> "OrderedDict[str, np.ndarray]":
    """Weights dict keyed by dump-dir relative names (no '.npy').

    Linear/Conv ~ N(0, gain^2/d_in); LayerNorm gamma = 1 + 0.1 N, beta = 0.02 N, eps = 1e-5;
    encoder positions = sinusoid table; decoder positions ~ N(0, 0.5^2); token embedding
    ~ N(0, logit_scale^2 / d) so the tied-embedding logits have std ~ logit_scale and greedy
    / beam decisions are separated by far more than fp32 round-off ("peaky" fixtures).
    Linear weights are stored [d_in, d_out] as the dump does (dump.py:141-145).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    d = dims["n_audio_state"]
    w: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def randn(*shape, scale=1.0):
        return (rng.standard_normal(shape, dtype=np.float32) * np.float32(scale))

    def scalar(v):
        return np.array([v], dtype=np.float32)

    # A plain 1/sqrt(d_in) random init degenerates: the residual stream is dominated by
    # context-independent offsets (biases, LayerNorm beta, the positive mean of GELU), so one
    # token wins whatever the audio or the history and every parity test would pass trivially.
    # The gains below keep the fixture honest: small biases, sharp attention (query gain) with
    # strong attention outputs so logits depend on WHICH encoder frames / past tokens are
    # attended, a conv stem gain so the audio outweighs the positional table, and sizeable
    # decoder positions so consecutive steps differ.
    bias_s, beta_s = 0.02, 0.02
    q_gain, self_out, cross_out, mlp_out, conv1_gain, dec_pos_s = 3.0, 3.0, 6.0, 1.0, 4.0, 0.5

    def linear(p, d_in, d_out, bias=True, gain=1.0):
        w[p + "/weight"] = randn(d_in, d_out, scale=gain / math.sqrt(d_in))
        if bias:
            w[p + "/bias"] = randn(d_out, scale=bias_s)

    def layer_norm(p, n):
        w[p + "/weight"] = 1.0 + randn(n, scale=0.1)
        w[p + "/bias"] = randn(n, scale=beta_s)
        w[p + "/eps"] = scalar(1e-5)

    def attention(p, n_head, out_gain):
        w[p + "/n_head"] = scalar(n_head)
        linear(p + "/query", d, d, gain=q_gain)
        linear(p + "/key", d, d, bias=False)        # mod.rs:402-404: key has no bias
        linear(p + "/value", d, d)
        linear(p + "/out", d, d, gain=out_gain)

    def mlp(p):
        linear(p + "/mlp1", d, 4 * d)
        linear(p + "/mlp2", 4 * d, d, gain=mlp_out)

    e = "encoder"
    w[e + "/n_layer"] = scalar(dims["n_audio_layer"])
    w[e + "/n_mels"] = scalar(dims["n_mels"])
    w[e + "/n_audio_state"] = scalar(d)
    w[e + "/positional_embedding"] = sinusoids(dims["n_audio_ctx"], d)
    w[e + "/conv1/weight"] = randn(d, dims["n_mels"], 3, scale=conv1_gain / math.sqrt(3 * dims["n_mels"]))
    w[e + "/conv1/bias"] = randn(d, scale=bias_s)
    w[e + "/conv2/weight"] = randn(d, d, 3, scale=1.0 / math.sqrt(3 * d))
    w[e + "/conv2/bias"] = randn(d, scale=bias_s)
    for i in range(dims["n_audio_layer"]):
        p = f"{e}/block_{i}"
        attention(p + "/attn", dims["n_audio_head"], self_out)
        layer_norm(p + "/attn_ln", d)
        mlp(p + "/mlp")
        layer_norm(p + "/mlp_ln", d)
    layer_norm(e + "/ln_post", d)

    t = "decoder"
    w[t + "/n_layer"] = scalar(dims["n_text_layer"])
    w[t + "/token_embedding/weight"] = randn(dims["n_vocab"], d, scale=logit_scale / math.sqrt(d))
    w[t + "/positional_embedding"] = randn(dims["n_text_ctx"], d, scale=dec_pos_s)
    for i in range(dims["n_text_layer"]):
        p = f"{t}/block_{i}"
        attention(p + "/attn", dims["n_text_head"], self_out)
        layer_norm(p + "/attn_ln", d)
        attention(p + "/cross_attn", dims["n_text_head"], cross_out)
        layer_norm(p + "/cross_attn_ln", d)
        mlp(p + "/mlp")
        layer_norm(p + "/mlp_ln", d)
    layer_norm(t + "/ln", d)
    return w


def synth_preset(name: str, logit_scale: float = 6.0):
    return synth_weights(preset_dims(name), preset_seed(name), logit_scale)


def micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=1031, n_audio_ctx=1500, n_text_ctx=448) -> dict:
    """A small model at the real head size (d_h = 64) for fast oracle runs."""
    return dict(n_mels=80, n_audio_ctx=n_audio_ctx, n_audio_state=n_state, n_audio_head=n_head,
                n_audio_layer=n_layer, n_vocab=n_vocab, n_text_ctx=n_text_ctx, n_text_state=n_state,
                n_text_head=n_head, n_text_layer=n_layer)


def synth_audio(n_samples: int, seed: int, sample_rate: int = 16000) -> np.ndarray:
    """16 kHz f32 in [-1, 1]: pink-ish noise at -30 dBFS (no bin sits on the 1e-10 floor),
    3-6 chirps / AM tones 100 Hz-7 kHz at -12 dBFS, and 200-500 ms gaps of -60 dBFS noise
    (never digital zero)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = int(n_samples)
    white = rng.standard_normal(n + 64)
    a = 0.97
    low = lfilter([1.0], [1.0, -a], white)          # one-pole low-pass, f64
    pink = 0.5 * white + 0.5 * low * math.sqrt(1 - a * a)
    pink = pink[64:]
    pink *= (10 ** (-30 / 20)) / (np.sqrt(np.mean(pink ** 2)) + 1e-30)
    t = np.arange(n, dtype=np.float64) / sample_rate
    sig = pink.copy()
    n_tones = int(rng.integers(3, 7))
    amp = 10 ** (-12 / 20) / n_tones
    for _ in range(n_tones):
        f0 = rng.uniform(100.0, 3000.0)
        f1 = rng.uniform(f0, 7000.0)
        dur = max(t[-1], 1e-3)
        phase = 2 * math.pi * (f0 * t + 0.5 * (f1 - f0) * t * t / dur)
        am = 0.6 + 0.4 * np.sin(2 * math.pi * rng.uniform(0.5, 4.0) * t + rng.uniform(0, 6.28))
        sig += amp * am * np.sin(phase + rng.uniform(0, 6.28))
    # gaps of -60 dBFS noise
    pos = 0
    while True:
        pos += int(rng.uniform(1.0, 4.0) * sample_rate)
        glen = int(rng.uniform(0.2, 0.5) * sample_rate)
        if pos + glen >= n:
            break
        sig[pos:pos + glen] = rng.standard_normal(glen) * 10 ** (-60 / 20)
        pos += glen
    return np.clip(sig, -1.0, 1.0).astype(np.float32)
