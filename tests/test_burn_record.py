"""The converted-model reader (`<name>.mpk.gz` + `<name>.cfg`; src/bin/convert/main.rs:17-19, :51 writes them,
src/bin/transcribe/main.rs:63-70, :116-126 loads them).  Burn 0.9.0 is not vendored with the reference, so the
fixtures are written by `tests/burnrecord.py` (test infrastructure) in the documented layout, in both plausible Param
nestings; format parity is unpinned (csrc/record_load.cpp header)."""
import gzip

import msgpack
import numpy as np
import pytest

import whisper_burn_amd as wb
import burnrecord
from whisper_burn_amd import synth


@pytest.fixture(scope="module")
def micro_weights():
    dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=1031)
    return synth.synth_weights(dims, seed=99)


@pytest.mark.parametrize("nesting,with_cfg", [("flat", True), ("data", False)])
def test_reader_returns_the_dump_tensors(tmp_path, micro_weights, nesting, with_cfg):
    w = micro_weights
    mpk, cfg = str(tmp_path / "m.mpk.gz"), str(tmp_path / "m.cfg")
    burnrecord.write_burn_record(w, mpk, cfg if with_cfg else None, nesting=nesting)
    got = wb.burn_record_tensors(mpk, cfg if with_cfg else None)
    assert "decoder/mask" not in got                        # mod.rs:125: stored, never needed (implicit causality)
    for name, ref in w.items():
        assert name in got, name
        ref = np.asarray(ref, dtype=np.float32)
        if ref.size == 1:                                   # scalars: n_layer, n_head, eps ...
            assert got[name].reshape(-1)[0] == ref.reshape(-1)[0], name
        else:
            assert got[name].shape == ref.shape, name
            assert np.array_equal(got[name], ref), name
    assert set(got) == set(w)


def test_reader_errors(tmp_path):
    with pytest.raises(wb.WbError) as e:
        wb.burn_record_tensors(str(tmp_path / "missing.mpk.gz"))
    assert e.value.status == -3
    p = str(tmp_path / "junk.mpk.gz")
    with gzip.open(p, "wb") as f:
        f.write(msgpack.packb({"metadata": {"float": "f32"}, "item": {"something": 1}}))
    with pytest.raises(wb.WbError) as e:
        wb.burn_record_tensors(p)
    assert e.value.status == -3 and "no Whisper module record" in str(e.value)
    p2 = str(tmp_path / "half.mpk.gz")
    with gzip.open(p2, "wb") as f:
        f.write(msgpack.packb({"metadata": {"float": "f16"}, "item": {}}))
    with pytest.raises(wb.WbError) as e:
        wb.burn_record_tensors(p2)
    assert "precision" in str(e.value)
    p3 = str(tmp_path / "trunc.mpk.gz")
    with gzip.open(p3, "wb") as f:
        f.write(msgpack.packb({"item": {"encoder": [1.0] * 100}}, use_single_float=True)[:-7])
    with pytest.raises(wb.WbError) as e:
        wb.burn_record_tensors(p3)
    assert e.value.status == -3


@pytest.mark.gpu
def test_model_from_record_equals_model_from_tensors(tmp_path, micro_weights):
    w = micro_weights
    mpk, cfg = str(tmp_path / "m.mpk.gz"), str(tmp_path / "m.cfg")
    burnrecord.write_burn_record(w, mpk, cfg)
    a, b = wb.Whisper.load_burn_record(mpk, cfg), wb.Whisper.from_tensors(w)
    assert a.dims == b.dims
    rng = np.random.default_rng(3)
    mel = rng.standard_normal((1, 80, 200)).astype(np.float32)
    toks = np.array([[5, 17, 300, 2]], dtype=np.int32)
    assert np.array_equal(a.forward(mel, toks), b.forward(mel, toks))
    a.close(); b.close()


def test_fixture_nesting_follows_the_reference_module_definitions(micro_weights):
    """The record fixture's nesting -- field names AND their order, including the stored `mask` Param and the usize fields
    that ride in the record -- is derived from the reference's own `#[derive(Module)]` structs (src/model/mod.rs:41-45,
    :117-126, :214-225, :286-292, :330-338, :366-371, :416-424 and the cross-attention struct).  Burn's named-MessagePack
    recorder serialises a module record field by field in declaration order; Burn itself is not vendored, so this is the
    closest in-tree pin of the layout the reader walks."""
    import os
    import re
    src = "/root/reference/src/model/mod.rs"
    if not os.path.exists(src):
        pytest.skip("the reference checkout is not present on this machine")
    text = open(src).read()
    structs = {m.group(1): re.findall(r"^\s*(\w+)\s*:", m.group(2), flags=re.M)
               for m in re.finditer(r"#\[derive\(Module, Debug\)\]\s*pub struct (\w+)<B: Backend>\s*\{(.*?)\n\}", text, flags=re.S)}
    rec = burnrecord.module_record(micro_weights)
    assert list(rec) == structs["Whisper"] == ["encoder", "decoder"]
    assert list(rec["encoder"]) == structs["AudioEncoder"]
    assert list(rec["decoder"]) == structs["TextDecoder"] and "mask" in structs["TextDecoder"]
    assert list(rec["encoder"]["blocks"][0]) == structs["ResidualEncoderAttentionBlock"]
    assert list(rec["decoder"]["blocks"][0]) == structs["ResidualDecoderAttentionBlock"]
    assert list(rec["encoder"]["blocks"][0]["attn"]) == structs["MultiHeadSelfAttention"]
    assert list(rec["decoder"]["blocks"][0]["cross_attn"]) == structs["MultiHeadCrossAttention"]
    assert list(rec["encoder"]["blocks"][0]["mlp"]) == structs["MLP"]
    # the usize fields are plain integers in the record, the mask a full [n_text_ctx, n_text_ctx] Param
    assert isinstance(rec["encoder"]["n_mels"], int) and isinstance(rec["decoder"]["n_text_ctx"], int)
    assert rec["decoder"]["mask"]["param"]["shape"] == [rec["decoder"]["n_text_ctx"]] * 2


def test_reader_error_names_the_layout_it_assumed(tmp_path):
    p = str(tmp_path / "junk.mpk.gz")
    with gzip.open(p, "wb") as f:
        f.write(msgpack.packb({"metadata": {"float": "f32"}, "item": {"something": 1}}))
    with pytest.raises(wb.WbError) as e:
        wb.burn_record_tensors(p)
    msg = str(e.value)
    assert "Burn 0.9.0" in msg and "not vendored" in msg and "mask, n_vocab, n_text_ctx" in msg and "{id, param:" in msg
