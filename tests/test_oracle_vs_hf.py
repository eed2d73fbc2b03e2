"""Pin the oracle against INDEPENDENT implementations of the same published model.

The reference has no tests / golden vectors, so the oracle is
cross-checked against Hugging Face `transformers` (code only, random-init, offline):
`WhisperFeatureExtractor` for the mel frontend and `WhisperModel` with the same
synthetic weights mapped in for the encoder / decoder.
"""
import numpy as np
import pytest
import torch

from oracle import mel as omel
from oracle.model import OracleWhisper
from whisper_burn_amd import synth

transformers = pytest.importorskip("transformers")


def test_mel_exact_twin_matches_hf_feature_extractor():
    from transformers import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor(feature_size=80, sampling_rate=16000, hop_length=160,
                                 chunk_length=30, n_fft=400)
    x = synth.synth_audio(480000, 77)
    hf = fe(x, sampling_rate=16000, return_tensors="np", padding="max_length")["input_features"][0]
    exact = omel.prep_audio_f64(x)
    assert hf.shape == exact.shape == (80, 3000)
    assert np.abs(hf - exact).max() < 5e-5


def test_mel_fp32_restatement_close_to_exact():
    x = synth.synth_audio(238559, 1236)
    o = omel.prep_audio(torch.from_numpy(x)[None])[0].numpy()
    e = omel.prep_audio_f64(x)
    assert o.shape == (80, 1490)
    assert np.abs(o - e).max() < 2e-4


def test_mel_filterbank_matches_f64():
    w32 = omel.get_mel_filters(16000.0).numpy()
    w64 = omel.mel_filters_f64(16000.0)
    assert w32.shape == (80, 201)
    assert np.abs(w32 - w64).max() < 1e-6
    assert (w32 != 0).sum(axis=1).min() >= 1


def _hf_model_from(weights, dims, n_pos):
    from transformers import WhisperConfig, WhisperModel
    d, H, nl = dims["n_audio_state"], dims["n_audio_head"], dims["n_audio_layer"]
    cfg = WhisperConfig(vocab_size=dims["n_vocab"], num_mel_bins=80, encoder_layers=nl,
                        encoder_attention_heads=H, decoder_layers=nl, decoder_attention_heads=H,
                        decoder_ffn_dim=4 * d, encoder_ffn_dim=4 * d, d_model=d,
                        max_source_positions=n_pos, max_target_positions=dims["n_text_ctx"],
                        activation_function="gelu", dropout=0.0, attention_dropout=0.0,
                        activation_dropout=0.0, scale_embedding=False, pad_token_id=0,
                        bos_token_id=1, eos_token_id=2, decoder_start_token_id=1,
                        attn_implementation="eager")
    m = WhisperModel(cfg).eval()
    sd = {}
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    W = weights

    def lin(dst, src, bias=True):
        sd[dst + ".weight"] = T(W[src + "/weight"].T)
        if bias:
            sd[dst + ".bias"] = T(W[src + "/bias"])

    def ln(dst, src):
        sd[dst + ".weight"] = T(W[src + "/weight"])
        sd[dst + ".bias"] = T(W[src + "/bias"])

    def attn(dst, src):
        lin(dst + ".q_proj", src + "/query")
        lin(dst + ".k_proj", src + "/key", bias=False)
        lin(dst + ".v_proj", src + "/value")
        lin(dst + ".out_proj", src + "/out")

    for c in ("conv1", "conv2"):
        sd[f"encoder.{c}.weight"] = T(W[f"encoder/{c}/weight"])
        sd[f"encoder.{c}.bias"] = T(W[f"encoder/{c}/bias"])
    sd["encoder.embed_positions.weight"] = T(W["encoder/positional_embedding"][:n_pos])
    for i in range(nl):
        attn(f"encoder.layers.{i}.self_attn", f"encoder/block_{i}/attn")
        ln(f"encoder.layers.{i}.self_attn_layer_norm", f"encoder/block_{i}/attn_ln")
        lin(f"encoder.layers.{i}.fc1", f"encoder/block_{i}/mlp/mlp1")
        lin(f"encoder.layers.{i}.fc2", f"encoder/block_{i}/mlp/mlp2")
        ln(f"encoder.layers.{i}.final_layer_norm", f"encoder/block_{i}/mlp_ln")
    ln("encoder.layer_norm", "encoder/ln_post")
    sd["decoder.embed_tokens.weight"] = T(W["decoder/token_embedding/weight"])
    sd["decoder.embed_positions.weight"] = T(W["decoder/positional_embedding"])
    for i in range(nl):
        attn(f"decoder.layers.{i}.self_attn", f"decoder/block_{i}/attn")
        ln(f"decoder.layers.{i}.self_attn_layer_norm", f"decoder/block_{i}/attn_ln")
        attn(f"decoder.layers.{i}.encoder_attn", f"decoder/block_{i}/cross_attn")
        ln(f"decoder.layers.{i}.encoder_attn_layer_norm", f"decoder/block_{i}/cross_attn_ln")
        lin(f"decoder.layers.{i}.fc1", f"decoder/block_{i}/mlp/mlp1")
        lin(f"decoder.layers.{i}.fc2", f"decoder/block_{i}/mlp/mlp2")
        ln(f"decoder.layers.{i}.final_layer_norm", f"decoder/block_{i}/mlp_ln")
    ln("decoder.layer_norm", "decoder/ln")
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(".k_proj.bias" in k for k in missing), missing
    return m


def test_encoder_decoder_match_hf_whisper_model():
    dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=1031)
    weights = synth.synth_weights(dims, seed=4242)
    T_frames, C = 600, 300
    hf = _hf_model_from(weights, dims, n_pos=C)
    rng = np.random.default_rng(0)
    mel = torch.from_numpy(rng.standard_normal((2, 80, T_frames)).astype(np.float32) * 0.5)
    tokens = torch.from_numpy(rng.integers(0, dims["n_vocab"], (2, 9)))
    # HF LayerNorm is sqrt(var + eps): compare against that oracle variant tightly ...
    o_in = OracleWhisper(weights, ln_eps_inside_sqrt=True)
    with torch.no_grad():
        enc_hf = hf.encoder(mel).last_hidden_state
        dec_hf = hf.decoder(input_ids=tokens, encoder_hidden_states=enc_hf).last_hidden_state
        logits_hf = dec_hf @ hf.decoder.embed_tokens.weight.T
    enc_o = o_in.forward_encoder(mel)
    logits_o = o_in.forward_decoder(tokens, enc_o)
    assert enc_o.shape == (2, C, 128)
    assert torch.allclose(enc_o, enc_hf, atol=2e-4, rtol=1e-4), (enc_o - enc_hf).abs().max()
    assert logits_o.shape == (2, 9, 1031)
    assert torch.allclose(logits_o, logits_hf, atol=2e-3, rtol=1e-4), (logits_o - logits_hf).abs().max()
    # ... and record that the Burn-0.9 variant (sqrt(var) + eps, the oracle default) differs
    # from it only at the O(eps / sigma) level.
    o_def = OracleWhisper(weights)
    delta = (o_def.forward(mel, tokens) - logits_o).abs().max().item()
    assert delta < 5e-3
    assert torch.equal(o_def.forward(mel, tokens).argmax(-1), logits_o.argmax(-1))
