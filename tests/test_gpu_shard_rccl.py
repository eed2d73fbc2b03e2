"""GPU: the C ABI's multi-GPU entry point with its built-in RCCL transport (csrc/shard.cpp), on the ONE GPU a test box
has: a world-size-1 communicator (ncclCommInitRank with one rank is legal) carries a real ncclAllGather over device staging
buffers, and wb_waveform_to_tokens_sharded reproduces the single-process path.  World sizes > 1 are covered on the CPU
(gloo, the real engine under the functional model: tests/test_emu_functional.py) and by the driver's scaling run."""
import numpy as np
import pytest

import whisper_burn_amd as wb
import workloads
from whisper_burn_amd import shard

pytestmark = pytest.mark.gpu


def test_rccl_transport_and_sharded_entry_point_world_1():
    comm = shard.RcclComm(shard.RcclComm.unique_id(), 0, 1, 0)
    payload = np.arange(4 * 109, dtype=np.int32).reshape(4, 109)
    assert np.array_equal(comm.allgather(payload), payload[None])            # ncclAllGather, one rank
    wl = workloads.WORKLOADS["tiny_bench"]
    eng = wb.Whisper.from_tensors(wl.weights())
    st = wb.SpecialTokens.for_vocab(eng.dims["n_vocab"])
    audio = wl.audio()
    params = wb.decode_params(st, 1, 24)
    ref, rwins = wb.waveform_to_tokens(eng, st, audio, 16000, params=params)
    got, wins = shard.waveform_to_tokens_sharded(eng, st, audio, 0, 1, comm=comm, params=params)
    assert got == ref and wins == rwins and len(wins) == 3
    comm.close()
    eng.close()
