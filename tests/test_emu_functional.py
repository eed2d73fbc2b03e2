"""CPU: the HIP sources themselves, executed on this GPU-less machine through the hipemu functional model
(whisper-burn_amd/tools/hipemu: fibers per thread, block barriers, the wave collectives and MFMA register layouts
by their ISA semantics), compared with the oracle at micro shapes -- plus the guard that keeps that build out of
the product path.  Not a substitute for the `-m gpu` parity tests (no timing, no memory model, micro shapes only);
it catches indexing / layout / launch-logic defects before a GPU minute is spent."""
import concurrent.futures
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "whisper-burn_amd")
EMU_DIR = os.path.join(PKG, "tools", "hipemu")
EMU_LIB = os.path.join(PKG, "lib", "libwhisper_hip_emu.so")
CLANG = os.environ.get("EMUCXX", "/opt/rocm/lib/llvm/bin/clang++")


@pytest.fixture(scope="module")
def emu_lib():
    if not (os.path.exists(CLANG) or shutil.which(CLANG)):
        pytest.skip("no host clang++ for the hipemu build")
    subprocess.run(["make", "-C", EMU_DIR, "-j", str(min(8, os.cpu_count() or 1))], check=True,
                   stdout=subprocess.DEVNULL)
    return EMU_LIB


@pytest.fixture(scope="module")
def emu_lib_prod(emu_lib):
    """The second functional-model library: the same sources with the PRODUCT's geometry constants (decode.h:
    CROSS_FUSED_MAX_C = 768 instead of the 384-key ring the micro-model library compiles)."""
    subprocess.run(["make", "-C", EMU_DIR, "-j", str(min(8, os.cpu_count() or 1)), "prod"], check=True,
                   stdout=subprocess.DEVNULL)
    return os.path.join(os.path.dirname(emu_lib), "libwhisper_hip_emu_prod.so")


def _spawn(emu_lib, which, extra_env=None, allow=True):
    env = dict(os.environ)
    env["WHISPER_HIP_LIB"] = emu_lib
    env.pop("WHISPER_HIP_ALLOW_EMU", None)
    if allow:
        env["WHISPER_HIP_ALLOW_EMU"] = "1"
    env["PYTHONPATH"] = os.pathsep.join([ROOT, PKG, os.path.join(ROOT, "tests"), env.get("PYTHONPATH", "")])
    env.setdefault("OMP_NUM_THREADS", "2")             # (several checks run side by side: the oracle half of each stays small)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "tests", "emu_checks.py"), which], env=env,
                          capture_output=True, text=True, timeout=900)


# The checks are independent processes, mostly single-threaded (the functional model runs one fiber at a time): the ones the
# selected tests will ask for are started ahead of time, a few side by side, and a test picks up its finished process.
_JOBS = {}
_POOL = concurrent.futures.ThreadPoolExecutor(max_workers=max(1, min(6, (os.cpu_count() or 2) - 2)))


def _key(which, extra_env, allow=True):
    return (which, tuple(sorted((extra_env or {}).items())), allow)


def _run(emu_lib, which, extra_env=None, allow=True):
    fut = _JOBS.pop(_key(which, extra_env, allow), None)
    return fut.result() if fut is not None else _spawn(emu_lib, which, extra_env, allow)


def _stamps_path(which, env):
    tag = "_".join([which] + [f"{k[-6:]}{v}" for k, v in sorted(env.items())])
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), f"ps_stamps_test_{tag}.bin")


def _persist_env(which, env):
    return dict(env, WHISPER_HIP_PS_STAMPS=_stamps_path(which, env))


# test function -> the (check, environment) jobs one parametrisation of it runs
_PLAN = {
    "test_kernel_sources_reproduce_the_oracle_under_the_functional_model": lambda p: [(p["which"], {})],
    "test_chained_greedy_windows_ending_at_different_steps": lambda p: [(p["which"], p["env"])],
    "test_persistent_flag_chained_decode_under_the_functional_model": lambda p: [(p["which"], _persist_env(p["which"], p["env"]))],
    "test_batch_mode_skinny_gemm_and_fused_streaming_blocks": lambda p: [(p["which"], p["env"])],
    "test_persistent_decode_falls_back_to_the_chain": lambda p: [("chain_eot", _persist_env("chain_eot_fb", p["env"]))],
    "test_two_pass_key_ring_of_the_fused_cross_attention": lambda p: [("geometry384", p["env"])],
    "test_split_precision_encoder_gemm_under_the_functional_model": lambda p: [
        ("forward", {"WHISPER_HIP_ENCODER_SPLIT": "1"}), ("greedy", {"WHISPER_HIP_ENCODER_SPLIT": "1"}), ("split_range", {}),
        ("forward", {"WHISPER_HIP_ENCODER_SPLIT": "0"}), ("greedy", {"WHISPER_HIP_ENCODER_SPLIT": "0"}),
        ("bitwise", {"WHISPER_HIP_ENCODER_SPLIT": "0"}), ("bitwise", {"WHISPER_HIP_ENCODER_SPLIT": "1"})],
    "test_the_other_kernel_template_families": lambda p: [(f"shape{p['d']}", {})],
    "test_unfused_decode_paths_under_the_functional_model": lambda p: [("greedy", {p["switch"]: "0"})],
    "test_sanitizer_and_reversed_schedule": lambda p: [(p["which"], {"HIPEMU_GUARD": "1", "HIPEMU_SEGV_TRACE": "1",
                                                                     "HIPEMU_ORDER": "reverse"})],
    "test_outputs_are_bitwise_independent_of_the_schedule": lambda p: [("bitwise", {}), ("bitwise", {"HIPEMU_ORDER": "reverse"})],
    "test_nine_to_sixteen_live_rows_on_the_fused_sublayer_path": lambda p: [("beam16", p["env"])],
    "test_decoder_split_range_guard_fails_loudly_and_falls_back": lambda p: [("dec_split_range", {})],
    "test_range_guard_words_are_per_session_and_the_encoder_check_is_deferred": lambda p: [(p["which"], {})],
}


@pytest.fixture(scope="module", autouse=True)
def _prefetch(request, emu_lib):
    for item in request.session.items:
        if getattr(item, "module", None) is not request.module:
            continue
        plan = _PLAN.get(getattr(item, "originalname", None) or item.name)
        if plan is None:
            continue
        cs = getattr(item, "callspec", None)
        for which, env in plan(dict(cs.params) if cs is not None else {}):
            k = _key(which, env)
            if k not in _JOBS:
                stamps = env.get("WHISPER_HIP_PS_STAMPS")
                if stamps and os.path.exists(stamps):
                    os.remove(stamps)
                _JOBS[k] = _POOL.submit(_spawn, emu_lib, which, env)
    yield
    for fut in _JOBS.values():
        fut.cancel()
    _JOBS.clear()


@pytest.mark.parametrize("which", ["mel", "greedy", "beam", "forward", "geometry", "prompted", "pool", "bigpad", "two_threads"])
def test_kernel_sources_reproduce_the_oracle_under_the_functional_model(emu_lib, which):
    p = _run(emu_lib, which)
    assert p.returncode == 0 and f"EMU_CHECK_OK {which}" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


@pytest.mark.parametrize("which,env", [("chain_eot", {}), ("chain_eot", {"WHISPER_HIP_POLL": "0"}),
                                       ("chain_eot_batch", {}),
                                       ("chain_eot_batch", {"WHISPER_HIP_CROSS_STREAM": "0", "WHISPER_HIP_BATCH_SKINNY": "0"})])
def test_chained_greedy_windows_ending_at_different_steps(emu_lib, which, env):
    """Dead rows of the device-chained greedy loop (a window that ended on <|endoftext|> while others go on), the
    last-finisher blanking of ST_N and the host's early exit, small-batch and batch mode."""
    p = _run(emu_lib, which, env)
    assert p.returncode == 0 and f"EMU_CHECK_OK {which}" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


@pytest.mark.parametrize("which,env", [("chain_eot", {"HIPEMU_CUS": "7"}), ("greedy", {}),
                                       ("persist384", {}), ("persist384x7", {"HIPEMU_CUS": "40"}),
                                       ("persist512", {"HIPEMU_CUS": "33", "HIPEMU_ORDER": "reverse"})])
def test_persistent_flag_chained_decode_under_the_functional_model(emu_lib, which, env):
    """decode_persist.hip: ONE co-resident grid runs every sublayer of every greedy step, blocks hand planes to each other
    through arrival counters (the functional model runs every block on its own thread and passes a baton whenever a block
    spins).  Token-exact against the oracle for d = 128 / 384 / 512, 4 and 7 rows, one role per block and several roles
    per block (HIPEMU_CUS shrinks the grid), both block orders.  It is the default greedy path of the models it supports."""
    env = _persist_env(which, env)
    p = _run(emu_lib, which, env)
    assert os.path.exists(env["WHISPER_HIP_PS_STAMPS"]), "the persistent kernel did not run"
    assert p.returncode == 0 and f"EMU_CHECK_OK {which}" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


@pytest.mark.parametrize("env", [{"WHISPER_HIP_PERSIST_INJECT_FAIL": "launch"}, {"HIPEMU_NO_COOP": "1"}],
                         ids=["launch-refused", "no-cooperative-launch"])  # (both prefill on the host, then run the chain)
def test_persistent_decode_falls_back_to_the_chain(emu_lib, env):
    """The persistent kernel is the default greedy path; when it cannot run -- the cooperative launch is refused (CU masking,
    a partition, a second cooperative client) or the device reports no cooperative launches at all (a wait that gives up
    before any step is committed takes the same exit) -- the session re-seeds the control block and decodes through the graph-replayed chain of
    one launch per sublayer instead of failing: same tokens (windows ending at different steps), no error."""
    env = _persist_env("chain_eot_fb", env)
    p = _run(emu_lib, "chain_eot", env)
    assert p.returncode == 0 and "EMU_CHECK_OK chain_eot" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
    # the role timeline is written only by a persistent launch that ran
    assert not os.path.exists(env["WHISPER_HIP_PS_STAMPS"]), "the persistent kernel ran although it was made to fail"


@pytest.mark.parametrize("which,env", [("beam_batch", {}), ("beam_batch", {"WHISPER_HIP_DECODER_SPLIT": "0"}),
                                       ("beam_batch", {"WHISPER_HIP_DECODER_SPLIT": "0", "WHISPER_HIP_SK_PAIR": "1"}),
                                       ("chain_eot_batch", {"WHISPER_HIP_CROSS_STREAM_FUSE": "0"})])
def test_batch_mode_skinny_gemm_and_fused_streaming_blocks(emu_lib, which, env):
    """decode_batch.hip under the functional model: the skinny split-K GEMM -- the split-precision default on
    v_mfma_f32_16x16x32_f16 (fp16 hi / lo weight tiles) and, with WHISPER_HIP_DECODER_SPLIT=0, the exact-f32 one on
    v_mfma_f32_16x16x4_f32 -- with 36 live rows
    (three row tiles; 9 windows x 4 beams, chunked cross-attention; also with the opt-in pairwise meeting of the waves'
    partial tiles), and the streaming cross-attention blocks without
    their fused front (fold + cross_attn_ln + Wq; the default at this width is fused, and the tiled GEMM the skinny one
    replaces runs too: test_chained_greedy_windows_ending_at_different_steps).  Token-exact against the oracle."""
    p = _run(emu_lib, which, env)
    assert p.returncode == 0 and f"EMU_CHECK_OK {which}" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


@pytest.mark.parametrize("env", [{}, {"WHISPER_HIP_FUSE16": "0"}], ids=["fused16", "batch"])
def test_nine_to_sixteen_live_rows_on_the_fused_sublayer_path(emu_lib, env):
    """beam 5 over three windows = 15 live rows (the reference's live setting on a 30 s chunk): the fused sublayer kernels with
    the MLP block and the logits GEMV as two row groups of 8, and the same rows through batch mode; d = 128 and d = 384;
    token-exact against the oracle's beam search (tests/emu_checks.py beam16)."""
    p = _run(emu_lib, "beam16", env)
    assert p.returncode == 0 and "EMU_CHECK_OK beam16" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_decoder_split_range_guard_fails_loudly_and_falls_back(emu_lib):
    """A decoder activation outside fp16's range under the split-precision skinny GEMM: the observing call returns
    WB_ERR_STATE, the model switches to the exact-f32 kernel, the retry decodes the rows of an engine that never used the split
    kernel (tests/emu_checks.py dec_split_range)."""
    p = _run(emu_lib, "dec_split_range", {})
    assert p.returncode == 0 and "EMU_CHECK_OK dec_split_range" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


@pytest.mark.parametrize("which", ["guard_chain", "guard_enc_deferred"])
def test_range_guard_words_are_per_session_and_the_encoder_check_is_deferred(emu_lib, which):
    """Round 6: the range-guard words belong to the session (no model-wide lock, no cross-session false trips).
    guard_chain: the DEVICE-CHAINED greedy path in batch mode with NaN rows -- the row's top-1 must not become an embedding
    index (it ends on <|endoftext|>), the call fails with WB_ERR_STATE, a bystander session of the same model is not failed,
    the retry is token-exact.  guard_enc_deferred: wb_waveform_to_tokens resolves the encoder's guard behind the decode's own
    synchronisation and decodes the batch again on the exact-f32 kernel (tests/emu_checks.py)."""
    p = _run(emu_lib, which, {})
    assert p.returncode == 0 and f"EMU_CHECK_OK {which}" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_device_beam_bookkeeping_under_the_functional_model(emu_lib):
    """tests/test_gpu_beam_device.py (the device restatement of beam.rs's bookkeeping against the host one on scripted log-prob
    rows with exact ties) runs unchanged over the functional model: the kernel's rank-based top-k, its wave ballots / 64-bit
    shuffles and the slot assignment are checked on the CPU too."""
    env = dict(os.environ)
    env["WHISPER_HIP_LIB"] = emu_lib
    env["WHISPER_HIP_ALLOW_EMU"] = "1"
    env["PYTHONPATH"] = os.pathsep.join([ROOT, PKG, os.path.join(ROOT, "tests"), env.get("PYTHONPATH", "")])
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_beam_device.py"), "-q", "-x",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900, cwd=os.path.join(ROOT, "tests"))
    assert p.returncode == 0 and " passed" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_production_key_ring_under_the_functional_model(emu_lib_prod):
    """The micro-model library compiles a 384-key ring so that small fixtures reach both of its code paths; the PRODUCT
    compiles 768 keys.  This runs the product's constant (`make prod`) at the real window lengths: C = 745 keys in one
    pass (the bench's geometry) and C = 1500 keys in two (the opt-in 30 s window), d = 384, persistent kernel and chain."""
    jobs = [("prodring", {}), ("prodring30", {"WHISPER_HIP_PERSIST": "0"}), ("prodring30", {})]
    futs = [_POOL.submit(_spawn, emu_lib_prod, which, env) for which, env in jobs]       # side by side
    for (which, env), f in zip(jobs, futs):
        p = f.result()
        assert p.returncode == 0 and f"EMU_CHECK_OK {which}" in p.stdout, (which, env, p.stdout[-2000:] + p.stderr[-4000:])


@pytest.mark.parametrize("env", [{}, {"WHISPER_HIP_PERSIST": "0"}])
def test_two_pass_key_ring_of_the_fused_cross_attention(emu_lib, env):
    """A window with more keys than one pass of the fused cross-attention block holds (the opt-in doubled window: C = 1500
    on the device against 768 per pass; here C = 395 against 384): K and V each make two passes through the register ring,
    in the persistent kernel and in the one-launch-per-sublayer chain.  Token-exact against the oracle."""
    p = _run(emu_lib, "geometry384", env)
    assert p.returncode == 0 and "EMU_CHECK_OK geometry384" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_split_precision_encoder_gemm_under_the_functional_model(emu_lib):
    """gemm_f16x3.hip (the default arithmetic of the encoder-side Linear layers since round 4; WHISPER_HIP_ENCODER_SPLIT=0
    selects the exact-f32 MFMA kernel): weights pre-split into fp16 hi / lo pieces at load, activations split on their way
    into LDS, three v_mfma_f32_32x32x16_f16 per product: logits of wb_forward within 1e-3 of the oracle, greedy tokens exact
    with EITHER kernel -- and the bits differ between the two (each kernel really ran).  `split_range`: an activation
    outside fp16's range trips the kernel's guard, the pass is repeated on the f32 kernel and the model stays there."""
    for which, v in (("forward", "1"), ("greedy", "1"), ("forward", "0"), ("greedy", "0")):
        p = _run(emu_lib, which, {"WHISPER_HIP_ENCODER_SPLIT": v})
        assert p.returncode == 0 and f"EMU_CHECK_OK {which}" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
    p = _run(emu_lib, "split_range", {})
    assert p.returncode == 0 and "EMU_CHECK_OK split_range" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
    digests = []
    for v in ("0", "1"):
        p = _run(emu_lib, "bitwise", {"WHISPER_HIP_ENCODER_SPLIT": v})
        assert p.returncode == 0, p.stderr[-2000:]
        digests.append([l for l in p.stdout.splitlines() if l.startswith("DIGEST ")][-1])
    assert digests[0] != digests[1]


@pytest.mark.parametrize("d", [384, 768])
def test_the_other_kernel_template_families(emu_lib, d):
    """d = 384 (the fused sublayer kernels of tiny.en; base.en's d = 512 instantiates the same templates) and 768
    (per-matrix GEMVs, `small`'s path); tools/emu_fuzz.py draws 512 as well."""
    p = _run(emu_lib, f"shape{d}")
    assert p.returncode == 0 and f"EMU_CHECK_OK shape{d}" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


@pytest.mark.parametrize("switch", ["WHISPER_HIP_FUSE_SUB", "WHISPER_HIP_FUSE_X", "WHISPER_HIP_CHAIN", "WHISPER_HIP_GRAPH", "WHISPER_HIP_PERSIST"])
def test_unfused_decode_paths_under_the_functional_model(emu_lib, switch):
    p = _run(emu_lib, "greedy", {switch: "0"})
    assert p.returncode == 0 and "EMU_CHECK_OK greedy" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


@pytest.mark.parametrize("which", ["mel", "greedy", "beam", "forward"])
def test_sanitizer_and_reversed_schedule(emu_lib, which):
    """HIPEMU_GUARD=1: every device allocation ends against an inaccessible page and fresh memory is 0xFF-poisoned (NaN /
    -1), so an out-of-bounds access faults and a read of never-written memory poisons the result.  HIPEMU_ORDER=reverse:
    blocks, waves and lanes run in descending order -- a kernel that needs a particular order is missing a barrier."""
    p = _run(emu_lib, which, {"HIPEMU_GUARD": "1", "HIPEMU_SEGV_TRACE": "1", "HIPEMU_ORDER": "reverse"})
    assert p.returncode == 0 and f"EMU_CHECK_OK {which}" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_two_ranks_shard_the_windows_of_the_real_engine(emu_lib):
    """The N > 1 path end to end on CPU: two gloo ranks, each decoding its block of windows with the engine itself
    (functional model), one all-gather of the token rows, host stitch -- equal to the oracle's single-process run."""
    p = _run(emu_lib, "sharded")
    assert p.returncode == 0 and "EMU_CHECK_OK sharded" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


@pytest.mark.parametrize("extra", [["--large-v2-leg", "on", "--large-v2-seconds", "4"], ["--geometry", "whisper30", "--beam", "2"],
                                   ["--encoder", "f32", "--large-v2-leg", "off", "--beam5-leg", "off"]],
                         ids=["default+large-v2-leg", "whisper30-beam2", "f32-encoder"])
def test_bench_main_runs_to_its_json_line(emu_lib, extra):
    """bench.py cannot start without cuda:0, and a broken bench line cannot be repaired after a round: tools/
    bench_dry_run.py stubs the torch.cuda calls and runs the script UNCHANGED over the functional model with a micro
    checkpoint -- every leg executes (timed steps, profiled passes, frontend, CPU baseline, JSON assembly)."""
    import json
    env = dict(os.environ)
    env["WHISPER_HIP_LIB"] = emu_lib
    p = subprocess.run([sys.executable, os.path.join(PKG, "tools", "bench_dry_run.py"), "--steps", "1", "--warmup", "0",
                        "--mel-windows", "2", "--seconds", "4", "--max-depth", "4"] + extra, env=env, capture_output=True,
                       text=True, timeout=600)
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    assert p.returncode == 0 and len(lines) == 1, p.stdout[-1500:] + p.stderr[-3000:]
    out = json.loads(lines[0])
    assert out["metric"].startswith("real-time factor") and out["value"] > 0 and out["n_gpus"] == 1 and out["steps"] == 1
    assert out["unit"] == "x real-time" and out["higher_is_better"] is True and out["scaling"] == "weak"
    assert "model" not in out["config"] and "workload" in out["config"] and out["vs_baseline"] is None
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["kernel"] == out["kernels"][0]["kernel"]
    assert 0 <= out["roofline"]["frac"] and out["roofline"]["peak"] == 8000.0 and out["roofline"]["unit"] == "GB/s"
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and set(cb["stages_s"]) == {"mel", "encoder", "decode", "total"}
    assert out["mel_frontend"]["windows"] >= 2 and out["stages"]["decode_launches_per_step"] > 0
    assert len(cb["runs_s"]) == 3
    assert out["config"]["tokens_checked"] is None and out["config"]["ranks_observed"]["world_size"] == 1     # (micro checkpoint: no golden)
    assert out["config"]["encoder_gemm"].startswith("exact-f32" if "--encoder" in extra else "split precision")
    assert cb["depth"] == 4 and cb["depth32"]["depth"] == 4 and cb["depth32"]["value"] > 0
    if "--beam" not in extra and "--beam5-leg" not in extra:   # the greedy reference-geometry line carries the live beam-5 setting too
        b5 = out["beam5"]
        assert b5["value"] > 0 and b5["steps"] == 2 and "beam_size 5" in b5["config"]["workload"]
    else:
        assert out["beam5"] is None
    if "--large-v2-seconds" in extra:      # the default tiny.en line carries the large-v2 leg at ONE GPU too (auto = on)
        lv = out["large_v2"]
        assert lv["n_gpus"] == 1 and lv["value"] > 0 and lv["steps"] == 5 and "large-v2" in lv["config"]["workload"]
        assert out["e2e_roofline"]["generated_tokens_per_window"] is not None
        assert lv["roofline"]["bound"] == "hbm" and lv["roofline"]["kernel"] == lv["kernels"][0]["kernel"]
        assert lv["stages"]["decode_ms_per_step_untraced"] > 0 and lv["stages"]["encoder_ms_per_step"] > 0
    else:
        assert out["large_v2"] is None


def test_bench_main_two_ranks_over_gloo(emu_lib):
    """The driver's N > 1 launch line (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) with the
    dry-run wrapper in place of bench.py: RCCL is swapped for gloo, everything else is the script's own N > 1 path (per-rank
    window blocks, the all-gather, MAX of the rank times, rank 0 prints ONE line with n_gpus = 2 and whole-job audio)."""
    import json
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["WHISPER_HIP_LIB"] = emu_lib
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(PKG, "tools", "bench_dry_run.py"), "--gpus", "2",
                        "--steps", "1", "--warmup", "0", "--mel-windows", "2", "--seconds", "8", "--max-depth", "4",
                        "--large-v2-seconds", "8", "--beam5-leg", "off"],
                       env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    assert p.returncode == 0 and len(lines) == 1, p.stdout[-1500:] + p.stderr[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["cpu_baseline"] is None      # CPU leg: N = 1 only
    assert out["config"]["windows"] >= 2                                                        # 16 s of audio in total: one window per rank
    assert abs(out["value"] - 16.0 / (out["ms_per_step"] * 1e-3)) < 0.05 * out["value"]           # whole-job audio / time
    lv = out["large_v2"]                                  # N > 1: the large-v2 leg runs by default (the 8-GPU headline config)
    assert lv["n_gpus"] == 2 and lv["config"]["windows"] >= 2
    assert abs(lv["value"] - 16.0 / (lv["ms_per_step"] * 1e-3)) < 0.05 * lv["value"]
    # the line is self-checking: the process group as torch.distributed reported it, every rank's (rank, local rank, device);
    # legs at N > 1 have no committed golden (the N-GPU clip is another signal): tokens_checked null, not true
    ro = out["config"]["ranks_observed"]
    assert ro["world_size"] == 2 and ro["backend"] == "gloo" and [r[0] for r in ro["rank_local_rank_device"]] == [0, 1]
    assert out["config"]["tokens_checked"] is None and lv["config"]["tokens_checked"] is None


def test_outputs_are_bitwise_independent_of_the_schedule(emu_lib):
    """split-K partial planes are folded in a fixed order and there are no float atomics, so results are
    bit-reproducible.  Under the functional model that claim is checkable: the SHA-256 of the raw output bytes (log-mel,
    stateless logits, top-k log-probs of forking beams, full log-prob rows, greedy tokens) is the same whether blocks,
    waves and lanes are scheduled in ascending or in descending order."""
    digests = []
    for order in ({}, {"HIPEMU_ORDER": "reverse"}):
        p = _run(emu_lib, "bitwise", order)
        assert p.returncode == 0 and "EMU_CHECK_OK bitwise" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
        digests.append([l for l in p.stdout.splitlines() if l.startswith("DIGEST ")][-1])
    assert digests[0] == digests[1], digests


def test_the_binding_refuses_the_functional_model_build(emu_lib):
    """The product path has no CPU route: _lib.load() raises on the hipemu build unless a test opts in."""
    p = _run(emu_lib, "greedy", allow=False)
    assert p.returncode != 0 and "hipemu functional-model build" in p.stderr
