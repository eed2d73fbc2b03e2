"""GPU: the DEVICE restatement of beam.rs's bookkeeping (decode.hip: dec_beam_update_kernel -- what wb_session_decode chains
between decode steps for beam_size > 1) against the HOST restatement (transcribe.cpp: beam_search_windows behind wb_beam_search,
itself pinned against oracle/beam.py by tests/test_host_logic.py), on SCRIPTED log-prob rows.

Real checkpoints almost never produce two equal f32 log-probs, so the decode tests cannot see whether the kernel's rank-based
top-k reproduces get_top_elements' tie rules (beam.rs:81-110: insert before the first stored score >= the new one, evict index
0), max_by's last-of-equals (beam.rs:23-27, :33-36) or the order in which finished beams are carried (beam.rs:56-57, :75-78).
The fake decoder of test_host_logic (log-prob row = f(window, token sequence), optionally QUANTISED: many exact ties, a small
vocabulary so that <|endoftext|> is often among the top k) drives both through the same wb_step_fn contract; token rows must be
identical for every window.
"""
import ctypes as C

import numpy as np
import pytest

import whisper_burn_amd as wb
from test_host_logic import FakeModel, _cpp_beam_search
from whisper_burn_amd import _lib

pytestmark = pytest.mark.gpu


def device_beam_search(model, params, n_windows):
    state = {"prev": []}

    def step(_user, new_tokens, parent, window, n, apply_mask, k, top_ids, top_lp):
        seqs = []
        for i in range(n):
            base = [] if parent[i] < 0 else state["prev"][parent[i]]
            seqs.append(base + [new_tokens[i]])
        for i in range(n):
            if k > 0:
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
    rc = _lib.load().wb_beam_search_device(0, C.byref(params), n_windows, model.V, C.cast(cb, C.c_void_p), None,
                                           toks.ctypes.data_as(_lib.c_int32_p), stride, lens.ctypes.data_as(_lib.c_int32_p))
    assert rc == 0, _lib.load().wb_last_error()
    return [toks[i, :lens[i]].tolist() for i in range(n_windows)]


def run_case(beam_size, quantum, n_windows, max_depth, seeds):
    V, n_special = 23, 4
    st = wb.SpecialTokens(start_of_transcript=V - 4, language=V - 3, transcribe=V - 2, no_timestamps=V - 1, end_of_text=V - 5,
                          is_special=np.array([0] * (V - n_special) + [1] * n_special, np.uint8))
    n_rows = 0
    for seed in seeds:
        model = FakeModel(V, n_special, seed, quantum)
        params = wb.decode_params(st, beam_size=beam_size, max_depth=max_depth)
        ref = _cpp_beam_search(model, params, n_windows)
        got = device_beam_search(model, params, n_windows)
        assert got == ref, (beam_size, quantum, n_windows, seed, got, ref)
        n_rows += len(ref)
    return n_rows


@pytest.mark.parametrize("quantum", [0.0, 0.5, 2.0])
@pytest.mark.parametrize("beam_size", [1, 2, 5, 8])
def test_device_beam_bookkeeping_matches_the_host_restatement(beam_size, quantum):
    assert run_case(beam_size, quantum, n_windows=3, max_depth=9, seeds=range(5)) == 15


def test_device_beam_bookkeeping_many_windows_in_several_passes():
    """More windows than the kernel's 16 waves: the window loop makes several passes; windows end at different depths."""
    assert run_case(5, 0.5, n_windows=37, max_depth=7, seeds=(11,)) == 37
    assert run_case(3, 2.0, n_windows=64, max_depth=5, seeds=(12,)) == 64
