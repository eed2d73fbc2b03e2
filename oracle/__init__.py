"""CPU oracle for the whisper-burn hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This package is a CPU *restatement* (PyTorch-CPU fp32, op for op in the
reference's order, plus a NumPy f64 "exact" twin of the mel frontend) of the
one hot path of Gadersd/whisper-burn:

    src/audio.rs        -> oracle/mel.py
    src/helper.rs       -> oracle/mel.py (tensor_max_scalar, tensor_log10, ...)
    src/model/mod.rs    -> oracle/model.py
    src/model/load.rs   -> (product) whisper-burn_amd/csrc/model_load.cpp; the Python dump-dir reader /
                           writer the tests use is whisper_burn_amd/dumpdir.py
    src/beam.rs         -> oracle/beam.py
    src/transcribe.rs   -> oracle/transcribe.py

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it, and only as the checker / the timed CPU baseline.
The product (`whisper-burn_amd/`) never imports it and has no CPU fallback.

PARITY UNPINNED.  The reference ships no tests, no golden vectors and no
benchmarks; it cannot be compiled here (no Rust toolchain, Burn 0.9.0 @
fb2a71bb and tch 0.13 are not vendored), and its single known-answer pair
(`audio.wav` -> `audio.txt`) needs Whisper weights and a tokenizer.json that do
not exist in this environment.  The arithmetic lives in the third-party crate
Burn 0.9.0 (git fb2a71bb81e1a688b4cdae38729b24dd9361283f, Cargo.lock:242-244)
on libtorch via tch 0.13.0 (Cargo.lock:3319-3320); its operator semantics are
restated from the published algorithm (see oracle/model.py header; the two
details that could not be verified against source -- LayerNorm epsilon
placement and max-subtraction in softmax -- are switchable).  What pins this
oracle instead are independent implementations of the same published model:
`transformers.WhisperFeatureExtractor` for the mel frontend and
`transformers.models.whisper.modeling_whisper` with the same synthetic weights
mapped in for the encoder/decoder (tests/test_oracle_vs_hf.py), plus
property tests of the host logic whose semantics are fully in-tree.
"""
