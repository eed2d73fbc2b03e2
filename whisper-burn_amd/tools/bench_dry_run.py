#!/usr/bin/env python3
"""Developer / CI tool: execute bench.py's main() on a machine WITHOUT a GPU.

The bench line is the one artefact of a round that cannot be fixed after the fact, and bench.py cannot start without
cuda:0 -- so this wrapper stubs the handful of torch.cuda calls the script makes (tensors stay on the CPU), points the
binding at the hipemu functional-model build of the engine (tools/hipemu) and swaps the model preset for a micro
checkpoint, then runs the script unchanged: every line of main() executes, including the profiled passes, the frontend
leg, the CPU baseline and the JSON assembly.  The numbers it prints are meaningless (an emulator on a CPU); the point is
that the script runs to its JSON line.  Test infrastructure only.

    python whisper-burn_amd/tools/bench_dry_run.py --steps 1 --warmup 0 --mel-windows 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
        whisper-burn_amd/tools/bench_dry_run.py --gpus 2 --steps 1 --warmup 0 --mel-windows 2      (the N > 1 path over gloo)
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "whisper-burn_amd")
os.environ.setdefault("WHISPER_HIP_LIB", os.path.join(PKG, "lib", "libwhisper_hip_emu.so"))
os.environ["WHISPER_HIP_ALLOW_EMU"] = "1"
os.environ.setdefault("WHISPER_BENCH_LARGE_WARMUP_S", "0")     # the large-v2 leg's settled warm-up: one step under the emulator
sys.path[:0] = [ROOT, PKG]

import torch  # noqa: E402

torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda *a, **k: None
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.current_device = lambda: 0
_to, _empty, _tensor = torch.Tensor.to, torch.empty, torch.tensor


def _is_cuda(x):
    return (isinstance(x, torch.device) and x.type == "cuda") or (isinstance(x, str) and x.startswith("cuda"))


def _to_cpu(self, *a, **k):
    return _to(self, *tuple("cpu" if _is_cuda(x) else x for x in a), **k)


def _empty_cpu(*a, **k):
    if _is_cuda(k.get("device")):
        k["device"] = "cpu"
    return _empty(*a, **k)


def _tensor_cpu(*a, **k):
    if _is_cuda(k.get("device")):
        k["device"] = "cpu"
    return _tensor(*a, **k)


torch.Tensor.to = _to_cpu
torch.empty = _empty_cpu
torch.tensor = _tensor_cpu

# N > 1 (launched under torch.distributed.run like the driver does): RCCL needs GPUs, gloo carries the same collectives
import torch.distributed as _dist  # noqa: E402

_init_pg = _dist.init_process_group


def _init_gloo(backend=None, **k):
    k.pop("device_id", None)
    return _init_pg(backend="gloo", **k)


_dist.init_process_group = _init_gloo

from whisper_burn_amd import synth  # noqa: E402

synth.synth_preset = lambda name, **kw: synth.synth_weights(
    synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=1031), seed=4242)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
