"""The named parity workloads: BASELINE.json's configs as concrete seeded inputs.

Shared by tests/golden/make_golden.py (which freezes the oracle's LITERAL decode of each one into
tests/golden/oracle_outputs.npz) and by the CPU / GPU tests that compare against those rows.
Audio seeds are chosen so that the oracle's smallest top-2 logit gap over the whole decode is well above
fp32 round-off (asserted in tests/test_fixture_quality.py): a parity failure is then a defect, not a tie.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Tuple

from whisper_burn_amd import synth


@dataclass(frozen=True)
class Workload:
    model: str
    n_samples: int
    audio_seed: int
    beam: int
    depth: int
    windows: Optional[Tuple[int, ...]] = None      # None = every reference window
    recipe: Tuple[Tuple[str, float], ...] = ()      # synth_weights overrides
    # opt-in Whisper geometry (wb_model_set_frame_limit): windows of 2 n_audio_ctx frames -- NOT reference behaviour
    frame_limit_x2: bool = False

    def audio(self):
        return synth.synth_audio(self.n_samples, self.audio_seed)

    def weights(self):
        return synth.synth_preset(self.model, **dict(self.recipe))


# Beam search has no length normalisation (beam.rs:9-37), so with an <|endoftext|> that can win it stops after
# ~10 tokens; the long beam workloads use checkpoints without the EOT ramp and run to max_depth.
NO_EOT = (("eot_beta", 0.0),)


WORKLOADS = {
    # config #2: tiny.en, one 30 s chunk (3 reference windows), greedy, depth 100 -- exactly bench.py's step
    "tiny_bench": Workload("tiny.en", 480000, synth.BENCH_AUDIO_SEED, 1, 100),
    # the reference's live decode setting (beam 5 x depth 100, transcribe.rs:232-233) on the same audio
    "tiny_beam5": Workload("tiny.en", 480000, synth.BENCH_AUDIO_SEED, 5, 100, None, NO_EOT),
    # config #3: base.en, 30 s, beam 5: to max_depth 32 (no EOT), and with windows ending on EOT
    "base_beam5": Workload("base.en", 480000, 1237, 5, 32, None, NO_EOT),
    "base_beam5_eot": Workload("base.en", 480000, 1237, 5, 100),
    # config #4: small (multilingual, V = 51 865), 10 minutes -> 51 windows; five windows spread over the clip vs the
    # oracle's LITERAL loop to depth 32 (the depth-100 evidence on all 51 windows is teacher-forced: test_gpu_batchmode.py)
    "small_10min": Workload("small", 9600000, 1238, 1, 32, (0, 12, 25, 38, 50)),
    # config #5: large-v2, one full 14.9 s window, literal loop to depth 16
    "large_window": Workload("large-v2", 238559, 1239, 1, 16),
    # bench.py's large-v2 leg (450 s per GPU = 38 windows = one GPU's share of config #5's hour, checkpoint without the EOT
    # ramp): the FIRST and LAST window of the leg's own audio, literal loop to the leg's depth 100 -- the rows bench.py
    # checks its leg against outside the timed region, and tests/test_gpu_batchmode.py its 38-row batch
    "large_leg": Workload("large-v2", 7200000, synth.BENCH_AUDIO_SEED + 5, 1, 100, (0, 37), NO_EOT),
    # config #2(b), the "perf geometry" of SURVEY 8d: ONE window of T = 2990 (+10 zero) frames, C = 1500 encoder
    # positions -- Whisper's own 30 s chunk, which the reference cannot run (mod.rs:236-241 bounds the FRAMES by
    # n_audio_ctx); opt-in on both sides (wb_model_set_frame_limit / OracleWhisper(frame_limit_x2=True))
    "tiny_whisper30": Workload("tiny.en", 478559, synth.BENCH_AUDIO_SEED, 1, 100, None, (), True),
}
