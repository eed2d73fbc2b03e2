import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "whisper-burn_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


def pytest_sessionstart(session):
    """A fresh checkout has no lib/libwhisper_hip.so (git-ignored): build it once (hipcc cross-compiles gfx950
    without a GPU; on the GPU box the built library travels with the snapshot and this is a no-op)."""
    lib = os.environ.get("WHISPER_HIP_LIB", os.path.join(PKG, "lib", "libwhisper_hip.so"))
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build()


# Order of the GPU suite (the driver runs `pytest -x -q -m gpu`): cheap kernel-parity files first, the multi-minute
# batch-mode sessions at `small` / large-v2 last -- so that one failure in a long test cannot blank the evidence of the
# ~100 short ones (round 4: the large-v2 file sorted second and `-x` stopped the run after four tests).
_GPU_ORDER = [
    "test_gpu_parity.py", "test_gpu_golden.py", "test_gpu_workloads.py", "test_gpu_edge.py", "test_gpu_guard.py", "test_gpu_concurrency.py", "test_gpu_beam_device.py", "test_gpu_session.py",
    "test_gpu_switches.py", "test_gpu_budget.py", "test_gpu_e2e.py", "test_resample.py", "test_wav_ingest.py",
    "test_tokenizer_integration.py", "test_legacy_modes.py", "test_burn_record.py", "test_gpu_handoff.py",
    "test_gpu_shard_rccl.py", "test_gpu_scale.py", "test_gpu_batchmode.py",
]


def pytest_collection_modifyitems(config, items):
    rank = {name: i for i, name in enumerate(_GPU_ORDER)}

    def key(item):
        fname = os.path.basename(str(item.fspath))
        late = 1 if "large_v2" in item.name or "large_window" in item.name else 0     # within a file: large-v2 cases last
        return (rank.get(fname, len(_GPU_ORDER) - 2), late)

    items.sort(key=key)          # stable: the order inside a file is otherwise kept


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The measured parity distances of this run (tests/parity_log.py): printed here because `-q` keeps the terminal
    summary but drops the tests' own prints."""
    import parity_log
    if parity_log.RECORDS:
        terminalreporter.section("parity distances (hip vs oracle)", sep="-")
        for ln in parity_log.lines():
            terminalreporter.write_line(ln)
