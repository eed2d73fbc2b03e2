"""Oracle (test infrastructure): restatement of /root/reference/src/beam.rs.

Generic, literal restatement (`beam_search`, `beam_search_step`,
`get_top_elements`) plus a NumPy equivalent of the V-wide top-k scan
(`top_indices_fast`) that tests prove identical to the literal insertion scan,
so the oracle decode loop does not spend its time in a Python loop over 51 864
continuations per beam per step.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Sequence

import numpy as np


@dataclass
class BeamNode:
    """beam.rs:3-7."""
    seq: list = field(default_factory=list)
    log_prob: float = 0.0


def _max_by_log_prob(beams: Sequence[BeamNode]):
    """Rust `Iterator::max_by(|a, b| a.partial_cmp(b).unwrap())`: the LAST of
    several equally-maximum elements; NaN panics (beam.rs:23, :34)."""
    best = None
    for b in beams:
        if b.log_prob != b.log_prob:
            raise ValueError("partial_cmp(..).unwrap() on NaN (reference panics)")
        if best is None or b.log_prob >= best.log_prob:
            best = b
    return best


def get_top_elements(elems: Sequence, score: Callable, num: int) -> list:
    """beam.rs:81-110, literally: streaming insertion into an ascending list."""
    top: list = []
    scores: List[float] = []
    for e in elems:
        s = score(e)
        if len(top) == num:
            if s < scores[0]:
                continue
        idx = None
        for i, sc in enumerate(scores):
            if sc >= s:
                idx = i
                break
        if idx is not None:
            top.insert(idx, e)
            scores.insert(idx, s)
        else:
            top.append(e)
            scores.append(s)
        if len(top) > num:
            top.pop(0)
            scores.pop(0)
    return top


def top_indices_fast(scores: np.ndarray, num: int) -> np.ndarray:
    """Indices `get_top_elements(range(n), scores.__getitem__, num)` would return,
    in the same order: ascending score, ties in DESCENDING index; on ties at the
    eviction boundary the lower index survives. No NaNs allowed."""
    scores = np.asarray(scores, dtype=np.float64)
    assert not np.isnan(scores).any()
    n = scores.shape[0]
    order = np.lexsort((-np.arange(n), scores))   # score asc, then index desc
    return order[max(0, n - num):]


def beam_search_step(beams: List[BeamNode], next_fn: Callable, is_finished: Callable,
                     beam_size: int) -> List[BeamNode]:
    """beam.rs:39-79. `next_fn(beams)` -> per beam a list of (token, score)."""
    finished, new_beams = [], []
    continuations = next_fn(beams)
    for node, conts in zip(beams, continuations):
        if is_finished(node.seq):
            finished.append(node)
        else:
            for tok, lp in get_top_elements(conts, lambda c: c[1], beam_size):
                new_beams.append(BeamNode(seq=node.seq + [tok], log_prob=lp))
    return get_top_elements(new_beams, lambda b: b.log_prob, beam_size) + \
        get_top_elements(finished, lambda b: b.log_prob, beam_size)


def beam_search(initial_beams: List[BeamNode], next_fn: Callable, is_finished: Callable,
                beam_size: int, max_depth: int, step_fn: Callable = beam_search_step) -> list:
    """beam.rs:9-37."""
    beams = list(initial_beams)
    for _ in range(max_depth):
        best = _max_by_log_prob(beams)
        if best is not None and is_finished(best.seq):
            break
        beams = step_fn(beams, next_fn, is_finished, beam_size)
    best = _max_by_log_prob(beams)
    return best.seq if best is not None else []
