"""CPU study (not a test: the file name keeps pytest away): would error-compensated split precision on the 2.5 PF matrix
path keep the encoder inside the parity budget?  The round-2 review asked for the experiment: operands
A = A_hi + A_lo in a 16-bit format, three MFMAs hi.hi + hi.lo + lo.hi, f32 accumulation, for the encoder and cross-K/V GEMMs.

This file measures the NUMERICS half on the CPU, with the oracle as the model: every nn::Linear of the encoder (and the
decoder's cross-attention key / value projections, which run on the encoder output) is evaluated as the three-product sum
over operands rounded to the 16-bit format -- each product of two such values is exact in f32, the accumulation is f32, as on
the matrix cores -- and the result is compared with the f64 evaluation of the same algorithm:

    python tests/study_split_precision.py small|large-v2|tiny.en [n_windows]

  f32        the oracle as it is (CPU BLAS, blocked f32 sums)
  f16x3      hi = fp16(x), lo = fp16((x - hi) * 2^11); y = hi.hi + (hi.lo + lo.hi) * 2^-11      (22 mantissa bits)
  bf16x3     hi = bf16(x), lo = bf16(x - hi);          y = hi.hi + hi.lo + lo.hi               (16 mantissa bits)
  bf16x1     plain bf16 operands (the existing speed path's arithmetic)

Reported: max |encoder output - f64| and max |log-prob - f64| over the first decode positions of each window, next to the
largest operand magnitude (fp16 overflows at 65504).  DESIGN.md section 7 quotes the numbers."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "whisper-burn_amd"), os.path.join(ROOT, "tests")]
from oracle import mel as omel                 # noqa: E402
from oracle.model import OracleWhisper, log_softmax  # noqa: E402
from whisper_burn_amd import synth             # noqa: E402


def split(x: torch.Tensor, kind: str):
    if kind == "f16":
        hi = x.half().float()
        lo = ((x - hi) * 2048.0).half().float()
        return hi, lo, 1.0 / 2048.0
    hi = x.bfloat16().float()
    lo = (x - hi).bfloat16().float()
    return hi, lo, 1.0


class SplitOracle(OracleWhisper):
    """The oracle with the encoder-side linears evaluated in split precision."""

    def __init__(self, weights, kind):
        super().__init__(weights)
        self.kind = kind
        self.max_operand = 0.0
        self._wsplit = {}

    def _is_encoder_side(self, p: str) -> bool:
        return p.startswith("encoder/") or (p.startswith("decoder/") and "cross_attn" in p and p.endswith(("/key", "/value")))

    def linear(self, p, x):
        if not self._is_encoder_side(p):
            return super().linear(p, x)
        w = self.w[p + "/weight"]
        self.max_operand = max(self.max_operand, float(x.abs().max()), float(w.abs().max()))
        if self.kind == "bf16x1":
            y = x.bfloat16().float().matmul(w.bfloat16().float())
        else:
            fmt = "f16" if self.kind == "f16x3" else "bf16"
            if p not in self._wsplit:
                self._wsplit[p] = split(w, fmt)
            bh, bl, sb = self._wsplit[p]
            ah, al, sa = split(x, fmt)
            y = ah.matmul(bh) + (ah.matmul(bl) * sb + al.matmul(bh) * sa)
        b = self.w.get(p + "/bias")
        return y if b is None else y + b


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "small"
    n_win = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    torch.manual_seed(0)
    w = synth.synth_preset(model, eot_beta=0.0)
    audio = synth.synth_audio(238559 * n_win, 1240)
    o64 = OracleWhisper(w, dtype=torch.float64)
    variants = {"f32": OracleWhisper(w), "f16x3": SplitOracle(w, "f16x3"), "bf16x3": SplitOracle(w, "bf16x3"),
                "bf16x1": SplitOracle(w, "bf16x1")}
    V = o64.dims.n_vocab
    # a fixed token prefix (special tokens are the last ~1500 ids of the vocabulary: stay below them)
    rng = np.random.default_rng(7)
    toks = torch.from_numpy(rng.integers(100, V - 2000, size=(1, 12)).astype(np.int64))
    worst = {k: [0.0, 0.0] for k in variants}
    for wi in range(n_win):
        pcm = torch.from_numpy(audio[wi * 238559:(wi + 1) * 238559])[None]
        mel32 = omel.prep_audio(pcm)
        mel32 = torch.cat([mel32, torch.zeros(1, 80, 10)], 2)
        t0 = time.time()
        enc64 = o64.forward_encoder(mel32.double())
        lp64 = log_softmax(o64.forward_decoder(toks, enc64), -1)
        print(f"window {wi}: f64 twin {time.time() - t0:.0f} s, max |enc| {float(enc64.abs().max()):.2f}, "
              f"max |log-prob| {float(lp64.abs().max()):.1f}", flush=True)
        for name, o in variants.items():
            enc = o.forward_encoder(mel32)
            lp = log_softmax(o.forward_decoder(toks, enc), -1)
            de = float((enc.double() - enc64).abs().max())
            dl = float((lp.double() - lp64).abs().max())
            worst[name][0] = max(worst[name][0], de)
            worst[name][1] = max(worst[name][1], dl)
            mo = getattr(o, "max_operand", float("nan"))
            print(f"  {name:<7} encoder {de:.3e}   log-probs {dl:.3e}   max operand {mo:.1f}", flush=True)
    print(f"{model}, {n_win} window(s): worst over windows")
    for name, (de, dl) in worst.items():
        print(f"  {name:<7} encoder {de:.3e}   log-probs {dl:.3e}")


if __name__ == "__main__":
    main()
