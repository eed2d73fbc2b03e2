"""The reference's retired greedy decoder and its repetition detectors (transcribe.rs:314-447; dead code there, an
optional mode here): C ABI helpers against the oracle restatement on CPU, the KV-cached loop against the oracle's
full-re-run loop on the GPU."""
import numpy as np
import pytest
import torch

from oracle import transcribe as otr
from whisper_burn_amd import legacy


def _cases():
    rng = np.random.default_rng(3)
    out = [[], [1], [1, 1], [1, 2, 3, 4, 5] * 6, [7] * 23, [1, 2, 3] * 9 + [1, 2], list(range(40)),
           [9, 9, 1, 2, 3, 4, 5, 8, 1, 2, 3, 4, 5, 7, 7, 1, 2, 3, 4, 5, 6, 1, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5]]
    for _ in range(300):
        n = int(rng.integers(0, 60))
        alphabet = int(rng.integers(1, 4))
        seq = rng.integers(0, alphabet + 1, n).tolist()
        if rng.random() < 0.5 and n > 4:                       # plant a periodic tail
            p = int(rng.integers(1, 6))
            seq = seq[:n // 3] + (seq[:p] or [0]) * int(rng.integers(2, 8))
        out.append(seq)
    return out


def test_find_repeated_tokens_index_matches_the_oracle():
    hits = 0
    for seq in _cases():
        for ws, mr in ((5, 4), (1, 2), (2, 3), (3, 2)):
            ref = otr.find_repeated_tokens_index(seq, ws, mr)
            assert legacy.find_repeated_tokens_index(seq, ws, mr) == ref, (seq, ws, mr)
            hits += ref is not None
    assert hits > 50
    # the reference's own shape: five-token window seen four times before -> truncate at the second occurrence
    seq = [50, 1, 2, 3, 4, 5, 60, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 70, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5]
    assert legacy.find_repeated_tokens_index(seq, 5, 4) == (1, 7)
    with pytest.raises(Exception):
        legacy.find_repeated_tokens_index(seq, 5, 1)           # the reference would unwrap a missing second match


def test_repetition_period_and_first_repetition_end_match_the_oracle():
    found = 0
    for seq in _cases():
        for mr in (1, 2, 4):
            ref = otr.repetition_period(seq, mr)
            assert legacy.repetition_period(seq, mr) == ref, (seq, mr)
            if ref is not None:
                found += 1
                assert legacy.first_repetition_end(seq, ref) == otr.first_repetition_end(seq, ref), (seq, ref)
        for period in (0, 1, 2, 5):
            if len(seq) >= period:
                assert legacy.first_repetition_end(seq, period) == otr.first_repetition_end(seq, period), (seq, period)
    assert found > 50
    with pytest.raises(Exception):
        legacy.first_repetition_end([1, 2, 3], 4)              # usize underflow in the reference
    assert legacy.repetition_period([1, 2, 3] * 9, 4) == 3 and legacy.repetition_period(list(range(30)), 4) is None


@pytest.mark.gpu
def test_legacy_greedy_matches_the_oracle_loop():
    import whisper_burn_amd as wb
    from whisper_burn_amd import synth
    from oracle.model import OracleWhisper
    import parity_util as pu
    dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=512)
    stops = set()
    for seed, kw in ((4242, {}), (4243, {"eot_beta": 0.0}), (4244, {"succ_share": 0.6})):
        w = synth.synth_weights(dims, seed=seed, **kw)
        eng, o = wb.Whisper.from_tensors(w), OracleWhisper(w)
        st = wb.SpecialTokens.for_vocab(512)
        for aseed in (61, 62):
            audio = synth.synth_audio(16000 * 6, aseed)
            mel = wb.prep_audio(audio)[0]                          # the same log-mel on both sides
            got = legacy.legacy_greedy(eng, st, mel, max_tokens=48)
            ref = otr.legacy_greedy(o, pu.ost(st), torch.from_numpy(mel)[None], max_tokens=48)
            assert got == ref, (seed, aseed, got, ref)
            assert got[:4] == [st.start_of_transcript, st.language, st.transcribe, st.no_timestamps] and got[-1] == st.end_of_text
            stops.add("ctx" if len(got) == 49 else "early")
        eng.close()
    assert stops == {"ctx", "early"}, stops                       # both the length cap and an early stop were exercised


@pytest.mark.gpu
def test_prompt_conditioning_matches_the_oracle():
    """Optional mode: the prompt conditioning the reference wrote and disabled (transcribe.rs:43-50, :188-199, shadowed at
    :201).  Every window after the first starts from [<|startofprev|>, last five non-special tokens, the four-token
    prompt]; rows (prompt included) and the stitched stream equal the oracle's restatement, greedy and beam 3."""
    import whisper_burn_amd as wb
    from whisper_burn_amd import synth
    from oracle.model import OracleWhisper
    import parity_util as pu
    dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=1031)
    w = synth.synth_weights(dims, seed=4242)
    eng, o = wb.Whisper.from_tensors(w), OracleWhisper(w)
    st = wb.SpecialTokens.for_vocab(1031)
    audio = synth.synth_audio(16000 * 28, 13)                         # three reference windows
    plain, _ = wb.waveform_to_tokens(eng, st, audio, 16000, 1, 9)
    for beam in (1, 3):
        got, rows = legacy.waveform_to_tokens_prompted(eng, st, audio, 16000, beam_size=beam, max_depth=9)
        ref, rrows = otr.waveform_to_tokens(o, pu.ost(st), audio, 16000, beam, 9, return_windows=True,
                                            start_of_prev=st.start_of_prev)
        assert len(rows) == 3 and rows[0][:4] == [st.start_of_transcript, st.language, st.transcribe, st.no_timestamps]
        for r in rows[1:]:                                               # <|startofprev|> + 5 previous tokens + prompt
            assert r[0] == st.start_of_prev and r[6:10] == rows[0][:4] and not any(st.is_special[t] for t in r[1:6])
        assert rows == rrows and got == ref, (beam, rows, rrows)
        if beam == 1:
            assert got != plain                                          # the conditioning changes the transcript
    eng.close()
