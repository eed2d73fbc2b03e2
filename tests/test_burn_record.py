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
