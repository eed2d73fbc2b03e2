"""CPU tests of the C-ABI library's host side: it loads, exports every declared symbol, and its
restatement of the reference's L3 plumbing (windows, token-overlap stitch, beam search) agrees
with the Python restatement in oracle/ on random and adversarial (tie-heavy) inputs.
No compute entry point is called: there is no GPU here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st_

import whisper_burn_amd as wb
from oracle import beam as obeam
from oracle import mel as omel
from oracle import transcribe as otr
from whisper_burn_amd import _lib, dumpdir, shard, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "whisper_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(wb_[a-z_0-9]+)\s*\(", hdr)) - {"wb_step_fn"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()                      # binds every symbol or raises
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (wb_[a-z_0-9]+)", nm))
    assert declared <= exported, declared - exported
    assert lib.wb_version().startswith(b"whisper_hip")


def test_compute_entry_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(wb.WbError):
        wb.prep_audio(np.zeros((1, 1600), np.float32))


@given(st_.integers(min_value=0, max_value=5000))
def test_max_waveform_samples(n):
    assert wb.max_waveform_samples(n) == omel.max_waveform_samples(n)


def test_reference_window_geometry():
    assert wb.max_waveform_samples(1500 - 10) == 238559
    s, l = wb.window_extents(480000, 16000, 238559)
    assert s.tolist() == [0, 190559, 381118] and l.tolist() == [238559, 238559, 98882]
    assert len(wb.window_extents(9600000, 16000, 238559)[0]) == 51
    assert len(wb.window_extents(57600000, 16000, 238559)[0]) == 303


@settings(max_examples=200, deadline=None)
@given(st_.integers(0, 3_000_000), st_.sampled_from([8000, 16000, 22050]),
       st_.one_of(st_.integers(1, 50), st_.integers(20_000, 400_000)), st_.integers(0, 5))
def test_window_extents_match_oracle(n, sr, wlen, ov):
    if wlen < 20_000:
        n = n % 2000            # tiny windows: keep the window count small
    s, l = wb.window_extents(n, sr, wlen, ov)
    chunk_overlap = sr * ov
    shift = max(max(wlen - chunk_overlap, 0), 1)
    ref = [(i * shift, min(i * shift + wlen, n)) for i in range(max(n - 1, 0) // shift + 1)]
    assert [(int(a), int(a + b)) for a, b in zip(s, l)] == ref
    if ov == 3:
        assert ref == otr.window_extents(n, sr, wlen)


tok_lists = st_.lists(st_.integers(0, 4), min_size=0, max_size=60)


@settings(max_examples=300, deadline=None)
@given(tok_lists, tok_lists, st_.integers(0, 50), st_.integers(0, 6))
def test_find_chunk_overlap_matches_oracle(prev, curr, max_off, min_ov):
    assert wb.find_chunk_overlap(prev, curr, max_off, min_ov) == otr.find_chunk_overlap(prev, curr, max_off, min_ov)


@settings(max_examples=100, deadline=None)
@given(st_.lists(st_.lists(st_.integers(0, 3), min_size=0, max_size=30), min_size=0, max_size=6))
def test_stitch_windows_matches_oracle(windows):
    ref = []
    for w in windows:
        ref = otr.stitch(ref, w)
    rows = np.zeros((max(len(windows), 1), 32), np.int32)
    lens = np.zeros(max(len(windows), 1), np.int32)
    for i, w in enumerate(windows):
        rows[i, :len(w)] = w
        lens[i] = len(w)
    assert wb.stitch_windows(rows[:len(windows)], lens[:len(windows)]) == ref


@settings(max_examples=200, deadline=None)
@given(st_.lists(st_.floats(-3, 3, allow_nan=False, width=32), min_size=0, max_size=40), st_.integers(1, 7),
       st_.booleans())
def test_top_indices_fast_equals_literal_scan(scores, num, quantize):
    s = np.asarray(scores, dtype=np.float64)
    if quantize:
        s = np.round(s)                    # many exact ties
    lit = obeam.get_top_elements(list(range(len(s))), lambda i: s[i], num)
    assert obeam.top_indices_fast(s, num).tolist() == lit


# ---- beam search: C++ (wb_beam_search over a callback) vs oracle/beam.py ---------------------

class FakeModel:
    """Deterministic stand-in for the decoder: log-prob row = f(token sequence)."""

    def __init__(self, V, n_special, seed, quantum):
        self.V, self.seed, self.quantum = V, seed, quantum
        self.is_special = np.zeros(V, bool)
        self.is_special[V - n_special:] = True

    def row(self, seq, masked):
        h = hash((self.seed,) + tuple(int(t) for t in seq)) & 0xFFFFFFFF
        x = np.random.default_rng(h).standard_normal(self.V).astype(np.float32) * 2.0
        if self.quantum:
            x = (np.round(x / self.quantum) * self.quantum).astype(np.float32)
        if masked:
            x = np.where(self.is_special, -np.inf, x).astype(np.float32)
        x = x - x.max()
        return (x - np.log(np.exp(x).sum(dtype=np.float32))).astype(np.float32)


def _cpp_beam_search(model, params, n_windows):
    state = {"prev": []}

    def step(_user, new_tokens, parent, window, n, apply_mask, k, top_ids, top_lp):
        seqs = []
        for i in range(n):
            base = [] if parent[i] < 0 else state["prev"][parent[i]]
            seqs.append(base + [new_tokens[i]])
        for i in range(n):
            if k > 0:
                # the window index salts the fake model so windows decode differently
                lp = model.row([window[i]] + seqs[i], bool(apply_mask))
                order = np.lexsort((np.arange(model.V), -lp.astype(np.float64)))[:k]
                for j in range(k):
                    top_ids[i * k + j] = int(order[j])
                    top_lp[i * k + j] = float(lp[order[j]])
        state["prev"] = seqs
        return 0

    cb = _lib.STEP_FN(step)
    stride = 4 + params.max_depth + 2
    toks = np.zeros((n_windows, stride), np.int32)
    lens = np.zeros(n_windows, np.int32)
    rc = _lib.load().wb_beam_search(C.byref(params), n_windows, model.V, C.cast(cb, C.c_void_p), None,
                                    toks.ctypes.data_as(_lib.c_int32_p), stride,
                                    lens.ctypes.data_as(_lib.c_int32_p))
    assert rc == 0, _lib.load().wb_last_error()
    return [toks[i, :lens[i]].tolist() for i in range(n_windows)]


def _oracle_beam_search(model, st, window, beam_size, max_depth, eot):
    def next_fn(beams):
        max_len = max(len(b.seq) for b in beams)
        out = []
        for b in beams:
            lp = model.row([window] + [t for t, _ in b.seq], not max_len > 5).astype(np.float64)
            out.append([((t, lp[t]), b.log_prob + lp[t]) for t in range(model.V)])
        return out

    init = obeam.BeamNode(seq=[(t, 0.0) for t in st], log_prob=0.0)
    seq = obeam.beam_search([init], next_fn, lambda s: bool(s) and s[-1][0] == eot, beam_size, max_depth)
    return [t for t, _ in seq]


@pytest.mark.parametrize("quantum", [0.0, 0.5, 2.0])
@pytest.mark.parametrize("beam_size", [1, 2, 5])
def test_beam_search_matches_oracle(beam_size, quantum):
    V, n_special = 23, 4
    for seed in range(6):
        model = FakeModel(V, n_special, seed, quantum)
        st = wb.SpecialTokens(20, 21, 22, 22, 19, model.is_special.astype(np.uint8))
        params = wb.decode_params(st, beam_size=beam_size, max_depth=9)
        got = _cpp_beam_search(model, params, n_windows=3)
        for w in range(3):
            ref = _oracle_beam_search(model, (20, 21, 22, 22), w, beam_size, 9, 19)
            assert got[w] == ref, (seed, w)


def test_beam_search_stops_when_best_beam_finished():
    # EOT (not special here) made overwhelmingly likely: every window must stop right after it
    class Eager(FakeModel):
        def row(self, seq, masked):
            lp = np.full(self.V, -20.0, np.float32)
            lp[3] = -0.001
            return lp
    model = Eager(11, 0, 0, 0.0)
    st = wb.SpecialTokens(7, 8, 9, 10, 3, np.zeros(11, np.uint8))
    got = _cpp_beam_search(model, wb.decode_params(st, beam_size=5, max_depth=50), 2)
    assert got == [[7, 8, 9, 10, 3]] * 2


# ---- fixtures / formats -----------------------------------------------------------------------

def test_dump_dir_round_trip(tmp_path):
    dims = synth.micro_dims(n_state=64, n_head=1, n_layer=1, n_vocab=64, n_audio_ctx=20, n_text_ctx=16)
    w = synth.synth_weights(dims, seed=1)
    dumpdir.write_dump_dir(w, str(tmp_path))
    r = dumpdir.read_dump_dir(str(tmp_path))
    assert set(r) == set(w)
    for k in w:
        assert r[k].shape == np.asarray(w[k]).shape and np.array_equal(r[k], w[k]), k
    # the on-disk encoding: flat f32, dims prefix (dump.py:134-139), no key bias (dump.py:144)
    flat = np.load(tmp_path / "encoder" / "conv1" / "weight.npy")
    assert flat.ndim == 1 and flat.dtype == np.float32 and flat[:3].tolist() == [64.0, 80.0, 3.0]
    assert not (tmp_path / "encoder" / "block_0" / "attn" / "key" / "bias.npy").exists()


@given(st_.integers(0, 400), st_.integers(1, 16))
def test_partition_windows_is_a_contiguous_cover(n, world):
    blocks = [shard.partition_windows(n, r, world) for r in range(world)]
    assert blocks[0][0] == 0 and blocks[-1][1] == n
    for (a, b), (c, d) in zip(blocks, blocks[1:]):
        assert b == c and a <= b
    assert max(b - a for a, b in blocks) <= shard.rows_per_rank(n, world)


@given(st_.integers(0, 2000), st_.integers(1, 64))
def test_c_abi_partition_equals_the_python_one(n, world):
    """wb_shard_partition (what a Rust / C caller of the sharded entry point uses) = shard.partition_windows."""
    for r in sorted({0, world // 2, world - 1}):
        assert shard.c_partition_windows(n, r, world) == shard.partition_windows(n, r, world)


def test_sharded_entry_point_validates_its_arguments_before_touching_a_device():
    """wb_waveform_to_tokens_sharded / wb_shard_partition: argument errors are reported without a GPU (nothing is decoded)."""
    lib = _lib.load()
    lo, hi = C.c_int64(0), C.c_int64(0)
    assert lib.wb_shard_partition(10, 3, 3, C.byref(lo), C.byref(hi)) != 0          # rank outside [0, world)
    assert lib.wb_shard_partition(10, 0, 0, C.byref(lo), C.byref(hi)) != 0          # world < 1
    assert lib.wb_shard_partition(0, 0, 4, C.byref(lo), C.byref(hi)) == 0 and (lo.value, hi.value) == (0, 0)
    n_st = C.c_int64(0)
    # null model / buffers
    rc = lib.wb_waveform_to_tokens_sharded(None, None, 0, 0, 16000, None, None, 0, 1, None, None, None, 108, None, 0, None, 0,
                                           C.byref(n_st))
    assert rc != 0 and b"null argument" in lib.wb_last_error()


@given(st_.integers(1, 6_000_000), st_.integers(1, 9))
@settings(max_examples=200, deadline=None)
def test_rank_pcm_span_reproduces_the_rank_windows(n, world):
    """a rank keeps only the PCM span of its own windows; inside the span they are windows 0 .. hi - lo - 1 of
    the same window_extents() with the same lengths (the span may yield one extra overlap-tail window: the next rank's)."""
    wlen = 238559
    starts, lens = wb.window_extents(n, 16000, wlen)
    for r in range(world):
        lo, hi = shard.partition_windows(len(starts), r, world)
        b, e = shard.rank_pcm_span(starts, lens, lo, hi)
        if hi == lo:
            assert (b, e) == (0, 0)
            continue
        s2, l2 = wb.window_extents(e - b, 16000, wlen)
        assert hi - lo <= len(s2) <= hi - lo + 1
        assert [int(x) + b for x in s2[:hi - lo]] == [int(x) for x in starts[lo:hi]]
        assert [int(x) for x in l2[:hi - lo]] == [int(x) for x in lens[lo:hi]]


def test_frontend_entry_validates_its_arguments_before_touching_a_device():
    """wb_waveform_to_mels_dev: window / stride errors are reported without a GPU (pointers are never read)."""
    starts = np.array([0, 190559], dtype=np.int64)
    lens = np.array([238559, 238559], dtype=np.int64)
    n = 190559 + 238559
    with pytest.raises(wb.WbError) as e:                    # row stride below the 1500 padded frames
        wb.waveform_to_mels_dev(0x1000, n, starts, lens, 0x2000, 80 * 1500, 1496)
    assert e.value.status == -1
    with pytest.raises(wb.WbError) as e:                    # window past the end of the waveform
        wb.waveform_to_mels_dev(0x1000, n - 1, starts, lens, 0x2000, 80 * 1500, 1500)
    assert e.value.status == -1
    with pytest.raises(wb.WbError) as e:                    # a window shorter than n_fft (audio.rs:292)
        wb.waveform_to_mels_dev(0x1000, n, starts, np.array([238559, 399], dtype=np.int64), 0x2000, 80 * 1500, 1500)
    assert e.value.status == -2


def test_end_to_end_roofline_model_of_the_bench():
    """bench.e2e_roofline_ms: stage terms scale the way SURVEY.md 8(d) states them."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    dims = dict(n_text_state=384, n_text_layer=4, n_vocab=51864)
    lens = [238559, 238559, 98882]                          # the 30 s bench workload
    a = bench.e2e_roofline_ms(dims, lens, 103, "f32")
    b = bench.e2e_roofline_ms(dims, lens, 206, "f32")
    c = bench.e2e_roofline_ms(dims, lens, 103, "f32", encoder_peak="f16x3")
    assert abs(b["decode"] - 2 * a["decode"]) < 1e-9 and b["mel"] == a["mel"]
    assert c["encoder_and_cross_kv"] < a["encoder_and_cross_kv"] and c["decode"] == a["decode"]
    frames = sum(n // 160 for n in lens)
    assert abs(a["mel"] - 960.0 * frames / 8e12 * 1e3) < 1e-12
    # logits stream alone: 4 * V * d bytes per step
    assert a["decode"] > 103 * 4 * 51864 * 384 / 8e12 * 1e3
    assert abs(a["total"] - (a["mel"] + a["encoder_and_cross_kv"] + a["decode"])) < 1e-12
    assert abs(a["_work"]["decode_bytes"] / 8e12 * 1e3 - a["decode"]) < 1e-9


def test_mel_constant_tables_match_the_oracle():
    """The Hann window and the Slaney filterbank the kernel uses (built once on the host in the reference's f32 op
    order) against the oracle's restatement of audio.rs:67-143 / :272-278, element by element."""
    import ctypes as C

    import torch
    from oracle import mel as omel
    lib = _lib.load()
    hann = np.zeros(400, np.float32)
    filt = np.zeros((80, 201), np.float32)
    assert lib.wb_mel_constants(16000.0, hann.ctypes.data_as(_lib.c_float_p), filt.ctypes.data_as(_lib.c_float_p)) == 0
    ref_h = omel.hann_window(400).numpy()
    ref_f = omel.get_mel_filters(16000.0, 400, 80).numpy()
    assert ref_f.shape == (80, 201)
    assert np.abs(hann - ref_h).max() <= 2e-7                         # sin() of libm vs torch: a few ulp at most
    scale = np.abs(ref_f).max()
    assert np.abs(filt - ref_f).max() <= 4e-7 * scale, np.abs(filt - ref_f).max()
    # the sparsity pattern: a tap is zero in one table only if it is tiny in the other (triangle edges)
    assert np.abs(ref_f[(filt == 0)]).max() <= 4e-7 * scale and np.abs(filt[(ref_f == 0)]).max() <= 4e-7 * scale
    nz = (filt != 0).sum(1)
    assert nz.min() >= 1 and nz.max() <= 16 and 380 <= (filt != 0).sum() <= 400   # 391 non-zeros, 1-14 per row
    del C


def test_cpu_baseline_leg_of_the_bench_runs_without_a_gpu():
    """bench.run_cpu_baseline: the oracle on one bounded window, stages timed separately, no GPU involved."""
    import importlib.util
    import os
    from whisper_burn_amd import synth
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=1031, n_audio_ctx=400)
    w = synth.synth_weights(dims, seed=4242)
    st = wb.SpecialTokens.for_vocab(1031)
    audio = synth.synth_audio(16000 * 6, 3)
    wlen = wb.max_waveform_samples(400 - 10)
    r = bench.run_cpu_baseline(w, st, audio, 16000, wlen, 1, 5, "reference", "micro")
    assert r["kind"] == "port" and r["value"] > 0 and r["cores"] >= 1
    s = r["stages_s"]
    assert abs(s["mel"] + s["encoder"] + s["decode"] - s["total"]) < 2e-3 and s["decode"] > 0
    assert "1 window (3.9 s" in r["sample"] and "1 warm-up + 3 runs, median" in r["sample"]
    assert len(r["runs_s"]) == 3 and min(r["runs_s"]) <= s["total"] <= max(r["runs_s"])


def test_special_mask_length_is_checked_before_it_reaches_c():
    """A tokenizer whose vocabulary is smaller than the model's would make the C side read past the mask buffer
    (it reads n_vocab bytes); the reference panics on that shape mismatch (transcribe.rs:243-275)."""
    import types

    from whisper_burn_amd import model as wm
    fake = types.SimpleNamespace(dims={"n_vocab": 100})
    assert wm.special_mask_bytes(fake, np.zeros(100, np.uint8)).shape == (100,)
    for n in (99, 101, 0):
        with pytest.raises(wm.WbError) as ei:
            wm.special_mask_bytes(fake, np.zeros(n, np.uint8))
        assert ei.value.status == wm.WB_ERR_SHAPE
