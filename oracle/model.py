"""Oracle (test infrastructure): CPU fp32 restatement of the Whisper encoder-decoder.

Follows /root/reference/src/model/mod.rs in PyTorch-CPU fp32, op for op.
Weights are a dict {dump-dir relative name -> np.ndarray / torch.Tensor}, named
as /root/reference/src/model/load.rs reads them (e.g. "encoder/block_0/attn/query/weight").

Burn 0.9.0 (git fb2a71bb, not vendored) operator semantics restated here:
  nn::Linear      y = x @ W + b, W stored [d_in, d_out]  (dump.py:141-145 transposes)
  conv::Conv1d    cross-correlation, weight [c_out, c_in, k], zero padding 1
  nn::LayerNorm   biased variance over the last dim; (x - mu) / (sqrt(var) + eps)
                  at this Burn revision  [UNVERIFIED against source -> switch
                  `ln_eps_inside_sqrt`; later Burn releases use sqrt(var + eps)]
  nn::GELU        exact erf form
  softmax         exp(x - max) / sum  [max-subtraction UNVERIFIED, <= few ulp]
  log_softmax     (x - max) - log(sum(exp(x - max)))
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class Dims:
    n_mels: int
    n_audio_ctx: int
    n_audio_state: int
    n_audio_head: int
    n_audio_layer: int
    n_vocab: int
    n_text_ctx: int
    n_text_state: int
    n_text_head: int
    n_text_layer: int


class OracleWhisper:
    """mod.rs:41-71 `Whisper` restated; `ln_eps_inside_sqrt` selects the LayerNorm variant."""

    def __init__(self, weights: dict, ln_eps_inside_sqrt: bool = False, dtype=torch.float32,
                 frame_limit_x2: bool = False):
        self.w = {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in weights.items()}
        self.dtype = dtype
        self.ln_eps_inside_sqrt = ln_eps_inside_sqrt
        # NOT reference behaviour (opt-in, the checker side of wb_model_set_frame_limit): bound the encoder POSITIONS
        # by n_audio_ctx instead of the mel frames, i.e. Whisper's own 30 s window (T = 3000 -> C = 1500).  The
        # reference asserts T <= n_audio_ctx (mod.rs:236-241) and would panic on such a window.
        self.frame_limit_x2 = frame_limit_x2
        w = self.w
        pe = w["encoder/positional_embedding"]
        te = w["decoder/token_embedding/weight"]
        pd = w["decoder/positional_embedding"]
        self.dims = Dims(
            n_mels=int(w["encoder/n_mels"].item()),
            n_audio_ctx=pe.shape[0],
            n_audio_state=int(w["encoder/n_audio_state"].item()),
            n_audio_head=int(w["encoder/block_0/attn/n_head"].item()),
            n_audio_layer=int(w["encoder/n_layer"].item()),
            n_vocab=te.shape[0],
            n_text_ctx=pd.shape[0],
            n_text_state=pd.shape[1],
            n_text_head=int(w["decoder/block_0/attn/n_head"].item()),
            n_text_layer=int(w["decoder/n_layer"].item()),
        )
        assert self.dims.n_audio_state == self.dims.n_text_state     # mod.rs:27-32
        self.mask = attn_decoder_mask(self.dims.n_text_ctx).to(dtype)  # load.rs:270

    # -- primitive modules ---------------------------------------------------
    def linear(self, p: str, x: torch.Tensor) -> torch.Tensor:
        y = x.matmul(self.w[p + "/weight"])
        b = self.w.get(p + "/bias")
        return y if b is None else y + b

    def layer_norm(self, p: str, x: torch.Tensor) -> torch.Tensor:
        eps = float(self.w[p + "/eps"].item())
        mean = x.mean(-1, keepdim=True)
        var = ((x - mean) ** 2).mean(-1, keepdim=True)
        if self.ln_eps_inside_sqrt:
            xn = (x - mean) / torch.sqrt(var + eps)
        else:
            xn = (x - mean) / (torch.sqrt(var) + eps)
        return xn * self.w[p + "/weight"] + self.w[p + "/bias"]

    def mlp(self, p: str, x: torch.Tensor) -> torch.Tensor:
        """mod.rs:376-382."""
        return self.linear(p + "/mlp2", F.gelu(self.linear(p + "/mlp1", x)))

    def self_attention(self, p: str, x, mask, n_head):
        """mod.rs:428-436."""
        q = self.linear(p + "/query", x)
        k = self.linear(p + "/key", x)
        v = self.linear(p + "/value", x)
        return self.linear(p + "/out", qkv_attention(q, k, v, mask, n_head))

    def cross_attention(self, p: str, x, xa, n_head):
        """mod.rs:482-490: K/V of xa recomputed on every call."""
        q = self.linear(p + "/query", x)
        k = self.linear(p + "/key", xa)
        v = self.linear(p + "/value", xa)
        return self.linear(p + "/out", qkv_attention(q, k, v, None, n_head))

    # -- encoder -------------------------------------------------------------
    def forward_encoder(self, mel: torch.Tensor) -> torch.Tensor:
        """mod.rs:228-260: [B, 80, T] -> [B, C, d], C = (T - 1) // 2 + 1."""
        mel = mel.to(self.dtype)
        _, n_mels, n_ctx = mel.shape
        assert n_mels == self.dims.n_mels                    # mod.rs:231-235
        assert n_ctx <= self.encoder_ctx_size()              # mod.rs:236-241 (n_audio_ctx unless frame_limit_x2)
        w = self.w
        x = F.gelu(F.conv1d(mel, w["encoder/conv1/weight"], w["encoder/conv1/bias"], padding=1))
        x = F.gelu(F.conv1d(x, w["encoder/conv2/weight"], w["encoder/conv2/bias"], stride=2, padding=1))
        x = x.transpose(1, 2)
        k = x.shape[1]
        x = x + w["encoder/positional_embedding"][0:k][None]
        H = self.dims.n_audio_head
        for i in range(self.dims.n_audio_layer):
            p = f"encoder/block_{i}"
            x = x + self.self_attention(p + "/attn", self.layer_norm(p + "/attn_ln", x), None, H)
            x = x + self.mlp(p + "/mlp", self.layer_norm(p + "/mlp_ln", x))
        return self.layer_norm("encoder/ln_post", x)

    # -- decoder -------------------------------------------------------------
    def forward_decoder(self, tokens: torch.Tensor, xa: torch.Tensor) -> torch.Tensor:
        """mod.rs:131-157: tokens [n, L] int, xa [n, C, d] -> logits [n, L, V]. Stateless."""
        tokens = torch.as_tensor(tokens).long()
        n, L = tokens.shape
        assert L <= self.dims.n_text_ctx                      # mod.rs:134-139
        w = self.w
        emb = w["decoder/token_embedding/weight"]
        x = emb[tokens] + w["decoder/positional_embedding"][0:L][None]
        H = self.dims.n_text_head
        for i in range(self.dims.n_text_layer):
            p = f"decoder/block_{i}"
            x = x + self.self_attention(p + "/attn", self.layer_norm(p + "/attn_ln", x), self.mask, H)
            x = x + self.cross_attention(p + "/cross_attn", self.layer_norm(p + "/cross_attn_ln", x), xa, H)
            x = x + self.mlp(p + "/mlp", self.layer_norm(p + "/mlp_ln", x))
        x = self.layer_norm("decoder/ln", x)
        return x.matmul(emb.transpose(0, 1)[None])

    def forward(self, mel: torch.Tensor, tokens: torch.Tensor) -> torch.Tensor:
        """mod.rs:48-50."""
        return self.forward_decoder(tokens, self.forward_encoder(mel))

    def encoder_ctx_size(self) -> int:
        return self.dims.n_audio_ctx * (2 if self.frame_limit_x2 else 1)

    def decoder_ctx_size(self) -> int:
        return self.dims.n_text_ctx


def softmax(x: torch.Tensor, dim: int) -> torch.Tensor:
    x = x - x.max(dim, keepdim=True).values
    e = torch.exp(x)
    return e / e.sum(dim, keepdim=True)


def log_softmax(x: torch.Tensor, dim: int) -> torch.Tensor:
    x = x - x.max(dim, keepdim=True).values
    return x - torch.log(torch.exp(x).sum(dim, keepdim=True))


def qkv_attention(q, k, v, mask, n_head: int) -> torch.Tensor:
    """mod.rs:493-533: both q and k scaled by d_h^-0.25; scores materialised."""
    n_batch, n_qctx, n_state = q.shape
    n_ctx = k.shape[1]
    scale = float(np.float32((n_state / n_head) ** -0.25)) if q.dtype == torch.float32 \
        else (n_state / n_head) ** -0.25
    n_hstate = n_state // n_head
    q = q.reshape(n_batch, n_qctx, n_head, n_hstate).transpose(1, 2) * scale
    k = k.reshape(n_batch, n_ctx, n_head, n_hstate).transpose(1, 2).transpose(2, 3) * scale
    v = v.reshape(n_batch, n_ctx, n_head, n_hstate).transpose(1, 2)
    qk = q.matmul(k)
    if mask is not None:
        qk = qk + mask[0:n_qctx, 0:n_ctx][None, None]
    w = softmax(qk, 3)
    return w.matmul(v).transpose(1, 2).flatten(2, 3)


def attn_decoder_mask(seq_length: int) -> torch.Tensor:
    """mod.rs:535-544: -inf strictly above the diagonal."""
    mask = torch.zeros(seq_length, seq_length)
    mask.masked_fill_(torch.ones(seq_length, seq_length, dtype=torch.bool).triu(1), float("-inf"))
    return mask
