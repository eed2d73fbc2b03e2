"""Driven by tests/test_emu_functional.py in a subprocess with WHISPER_HIP_LIB = lib/libwhisper_hip_emu.so: the engine's
own HIP sources, compiled for the host against the hipemu functional model (whisper-burn_amd/tools/hipemu), compared
with the oracle at micro shapes.  This is a CHECK OF THE KERNEL SOURCES' LOGIC on a machine without a GPU (indexing,
wave collectives, MFMA operand layouts, the host-side launch logic); it says nothing about the gfx950 build's timing
or memory ordering -- the `-m gpu` tests remain the parity tests proper."""
import sys

import numpy as np
import torch

import parity_util as pu
import whisper_burn_amd as wb
from oracle import mel as omel
from oracle import transcribe as otr
from oracle.model import OracleWhisper
from whisper_burn_amd import _lib, synth


def _shard_worker(rank, world, port, q):
    """One rank of the N > 1 path of bench.py, with the REAL engine (under the functional model) decoding its block of
    windows: windows sharded by shard.partition_windows, one gloo all-gather of the token rows, host stitch."""
    import os
    import torch.distributed as dist
    from whisper_burn_amd import shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=1031)
        eng = wb.Whisper.from_tensors(synth.synth_weights(dims, seed=4242))
        st = wb.SpecialTokens.for_vocab(1031)
        a = synth.synth_audio(16000 * 52, 17)                             # 5 reference windows
        n_win = len(wb.window_extents(len(a), 16000, wb.max_waveform_samples(1490))[0])
        row = 4 + 8 + 4

        def decode_local(lo, hi):
            return wb.waveform_to_tokens(eng, st, a, 16000, 1, 8, win_begin=lo, win_end=hi)[1]

        toks, wins = shard.transcribe_sharded(decode_local, wb.stitch_windows, n_win, rank, world, row)
        # ... and the same path behind the C ABI (csrc/shard.cpp: wb_waveform_to_tokens_sharded), the exchange handed in as a
        # callback -- here gloo's all-gather; on GPUs the library's own RCCL transport (wb_comm_allgather)
        import torch as _t

        def gather(local):
            t = _t.from_numpy(np.ascontiguousarray(local))
            out = _t.empty(world * t.numel(), dtype=t.dtype)
            dist.all_gather_into_tensor(out, t)
            return out.numpy().reshape(world, -1)

        ctoks, cwins = shard.waveform_to_tokens_sharded(eng, st, a, rank, world, allgather=gather,
                                                        params=wb.decode_params(st, 1, 8))
        assert ctoks == toks and cwins == wins, (rank, ctoks, toks)
        assert shard.c_partition_windows(n_win, rank, world) == shard.partition_windows(n_win, rank, world)
        q.put((rank, n_win, toks, wins))
        eng.close()
    finally:
        dist.destroy_process_group()


def main(which):
    assert b"hipemu" in _lib.load().wb_version()
    dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=1031)
    w = synth.synth_weights(dims, seed=4242)
    eng, o = wb.Whisper.from_tensors(w), OracleWhisper(w)
    st = wb.SpecialTokens.for_vocab(1031)
    if which == "mel":
        for n, seed in ((400, 1), (401, 2), (16000 * 2 + 77, 3)):
            a = synth.synth_audio(n, seed)
            got = wb.prep_audio(a[None])[0]
            ref = omel.prep_audio(torch.from_numpy(a)[None])[0].numpy()
            assert got.shape == ref.shape and np.abs(got - ref).max() < 2e-4, (n, np.abs(got - ref).max())
    elif which == "greedy":
        a = synth.synth_audio(16000 * 3, 7)
        got, wins = wb.waveform_to_tokens(eng, st, a, 16000, 1, 10)
        ref, rw = otr.waveform_to_tokens(o, pu.ost(st), a, 16000, 1, 10, return_windows=True)
        assert got == ref and wins == rw, (got, ref)
        assert len(set(got[4:])) >= 4                                   # not a degenerate sequence
    elif which in ("chain_eot", "chain_eot_batch"):
        # device-chained greedy decode whose windows end on <|endoftext|> at DIFFERENT steps: rows that have ended are
        # marked dead in the step state (their attention blocks exit, a finished window streams no cached K/V, its
        # token is frozen), the last finisher blanks ST_N while sibling blocks may not have started, and the host's
        # flag wait ends through its "every window finished" branch.  4 windows: the fused small-batch kernels;
        # 13 windows: batch mode (streaming cross-attention; WHISPER_HIP_CROSS_STREAM=0: chunked + combine).
        n_s, seed = (16000 * 40, 23) if which == "chain_eot" else (16000 * 150, 29)
        a = synth.synth_audio(n_s, seed)
        got, wins = wb.waveform_to_tokens(eng, st, a, 16000, 1, 30)
        ref, rw = otr.waveform_to_tokens(o, pu.ost(st), a, 16000, 1, 30, return_windows=True)
        gen = [len(r) - 4 for r in rw]
        assert wins == rw and got == ref, (wins, rw)
        assert len(set(gen)) >= 3 and all(r[-1] == st.end_of_text for r in rw) and max(gen) < 30, gen
    elif which == "two_threads":
        # calls from different host threads on one model: the sessions' calls and a stateless one at the same time, several
        # rounds; every result equals the single-threaded one (the process-wide GPU turn -- wb_internal.h: GpuTurn -- is
        # recursive: wb_waveform_to_tokens -> wb_session_decode; tests/test_gpu_concurrency.py is the GPU twin)
        import threading
        clips = [synth.synth_audio(16000 * 3, 7), synth.synth_audio(16000 * 4 + 123, 8)]
        ref = [wb.waveform_to_tokens(eng, st, c, 16000, 1, 6) for c in clips]
        mel_ref = wb.prep_audio(clips[0][None])[0]
        for rnd in range(2):
            out, err = [None] * 3, []

            def call(i):
                try:
                    out[i] = wb.waveform_to_tokens(eng, st, clips[i], 16000, 1, 6) if i < 2 else wb.prep_audio(clips[0][None])[0]
                except Exception as e:      # noqa: BLE001
                    err.append(e)

            th = [threading.Thread(target=call, args=(i,)) for i in range(3)]
            [t.start() for t in th]
            [t.join() for t in th]
            assert not err, err
            assert out[0] == ref[0] and out[1] == ref[1] and np.array_equal(out[2], mel_ref), rnd
    elif which == "pool":
        # a session that outlives its model: model A is freed, model B (same architecture, other weights) is loaded --
        # its handle may reuse A's heap address -- then A's leftover session is released and B decodes.  The session
        # pool is keyed by a never-reused model id, so A's session (captured graphs over A's freed weights) must not be
        # handed to B.
        a = synth.synth_audio(16000 * 3, 7)
        starts, lens = wb.window_extents(len(a), 16000, wb.max_waveform_samples(1490))
        for trial in range(4):
            ea = wb.Whisper.from_tensors(w)
            sa = wb.Session.begin(ea, a, starts, lens, max_beams=1)
            sa.set_special_mask(st.is_special)
            sa.decode(wb.decode_params(st, 1, 6))             # captures the decode graphs over A's buffers
            ea.close()                                        # model first ...
            w2 = synth.synth_weights(dims, seed=777 + trial)
            eb, ob = wb.Whisper.from_tensors(w2), OracleWhisper(w2)
            sa.close()                                        # ... then its session
            got, _ = wb.waveform_to_tokens(eb, st, a, 16000, 1, 6)
            assert got == otr.waveform_to_tokens(ob, pu.ost(st), a, 16000, 1, 6), trial
            eb.close()
    elif which == "bigpad":
        # padding above 1536 frames with short windows (allowed below max_mel_frames = 3000 with the opt-in geometry):
        # the mel grid is wider than the per-window (max, min) table, whose padding-only tiles must not be written
        e2, o2 = wb.Whisper.from_tensors(w), OracleWhisper(w, frame_limit_x2=True)
        e2.set_frame_limit(True)
        pad = 1700
        a = synth.synth_audio(16000 * 5, 43)
        starts, lens = np.array([0, 30000], np.int64), np.array([24000, 50000], np.int64)
        sess = wb.Session.begin(e2, a, starts, lens, max_beams=1, padding=pad)
        sess.set_special_mask(st.is_special)
        rows = sess.decode(wb.decode_params(st, 1, 6, padding=pad))
        for wi in range(2):
            mel = omel.prep_audio(torch.from_numpy(a[starts[wi]:starts[wi] + lens[wi]])[None])
            ref = otr.mels_to_tokens(o2, pu.ost(st), mel, pad, 1, 6)
            assert rows[wi] == ref, (wi, rows[wi], ref)
        sess.close(); e2.close()
    elif which.startswith("persist"):
        # the persistent flag-chained decode kernel at the other template families: d = 384 (tiny.en's) with 4 rows (MR = 4)
        # and 7 rows (MR = 8), d = 512 (base.en's) with 4 rows; two layers, windows of different length that end on
        # <|endoftext|> at different steps.  HIPEMU_CUS (set by the test) forces several roles per block.
        dd, n_s = {"persist384": (384, 50000), "persist384x7": (384, 96000), "persist512": (512, 50000)}[which]
        dims = synth.micro_dims(n_state=dd, n_head=dd // 64, n_layer=2, n_vocab=2053, n_audio_ctx=400)
        w2 = synth.synth_weights(dims, seed=90 + dd)
        e2, o2 = wb.Whisper.from_tensors(w2), OracleWhisper(w2)
        s2 = wb.SpecialTokens.for_vocab(2053)
        a = synth.synth_audio(n_s, 51)
        got, wins = wb.waveform_to_tokens(e2, s2, a, 16000, 1, 12)
        ref, rw = otr.waveform_to_tokens(o2, pu.ost(s2), a, 16000, 1, 12, return_windows=True)
        assert wins == rw and got == ref, (which, wins, rw)
        assert len(wins) == (7 if which.endswith("x7") else 4)
        e2.close()
    elif which == "prodring":
        # run with lib/libwhisper_hip_emu_prod.so (the product's key ring: CROSS_FUSED_MAX_C = 768): reference-length
        # windows (C = 745 keys: ONE pass through the ring, the bench's geometry) at d = 384 with two heads' worth of blocks
        # per row, through the persistent kernel and -- WHISPER_HIP_PERSIST=0 -- the one-launch-per-sublayer chain
        assert b"prod" in _lib.load().wb_version(), _lib.load().wb_version()
        dims = synth.micro_dims(n_state=384, n_head=6, n_layer=1, n_vocab=2053)      # n_audio_ctx = 1500: the real window
        w2 = synth.synth_weights(dims, seed=61)
        e2, o2 = wb.Whisper.from_tensors(w2), OracleWhisper(w2)
        s2 = wb.SpecialTokens.for_vocab(2053)
        a = synth.synth_audio(16000 * 17, 47)                                            # 2 windows: 745 and 260 keys
        got, wins = wb.waveform_to_tokens(e2, s2, a, 16000, 1, 6)
        ref, rw = otr.waveform_to_tokens(o2, pu.ost(s2), a, 16000, 1, 6, return_windows=True)
        assert got == ref and wins == rw and len(wins) == 2, (got, ref, wins, rw)
        e2.close()
    elif which == "prodring30":
        # ... and the opt-in 30 s window (C = 1500 keys: TWO passes through the 768-key ring)
        assert b"prod" in _lib.load().wb_version(), _lib.load().wb_version()
        dims = synth.micro_dims(n_state=384, n_head=6, n_layer=1, n_vocab=2053)
        w2 = synth.synth_weights(dims, seed=63)
        e2, o2 = wb.Whisper.from_tensors(w2), OracleWhisper(w2, frame_limit_x2=True)
        e2.set_frame_limit(True)
        s2 = wb.SpecialTokens.for_vocab(2053)
        a = synth.synth_audio(16000 * 31, 49)                                            # one 29.9 s window + its tail
        got, wins = wb.waveform_to_tokens(e2, s2, a, 16000, 1, 5)
        ref, rw = otr.waveform_to_tokens(o2, pu.ost(s2), a, 16000, 1, 5, return_windows=True)
        assert got == ref and wins == rw and len(wins) == 2, (got, ref, wins, rw)
        e2.close()
    elif which == "split_range":
        # range guard of the split-precision encoder GEMM (gemm_f16x3.hip): an activation outside fp16's range (here the
        # GELU output that feeds the first block's second MLP matrix, pushed to ~1e5 by its bias) makes an fp16 piece inf;
        # the kernel raises its flag, the pass is repeated on the exact-f32 kernel and the model stays there
        w2 = dict(w)
        w2["encoder/block_0/mlp/mlp1/bias"] = np.asarray(w["encoder/block_0/mlp/mlp1/bias"], np.float32) + np.float32(1.0e5)
        e2, o2 = wb.Whisper.from_tensors(w2), OracleWhisper(w2)
        assert e2.encoder_gemm() == "f16x3", e2.encoder_gemm()
        a = synth.synth_audio(16000, 5)
        mel = np.concatenate([wb.prep_audio(a[None]), np.zeros((1, 80, 10), np.float32)], 2)
        got = e2.forward_encoder(mel)
        assert e2.encoder_gemm() == "f32", e2.encoder_gemm()
        # the repeated pass IS the exact-f32 pass: a second call (now on the f32 kernel from the start) gives the same bits;
        # against the oracle only loosely -- activations of 1e5 in f32 leave ~1e-2 of the normalised output to cancellation
        assert np.isfinite(got).all() and np.array_equal(got, e2.forward_encoder(mel))
        ref = o2.forward_encoder(torch.from_numpy(mel)).numpy()
        assert np.abs(got - ref).max() < 0.3, np.abs(got - ref).max()
        # an ordinary model is untouched by the guard
        assert eng.encoder_gemm() == "f16x3"
        eng.forward_encoder(mel)
        assert eng.encoder_gemm() == "f16x3"
        e2.close()
    elif which == "dec_split_range":
        # range guard of the split-precision DECODER GEMM (decode_batch.hip, dec_skinny_f16x3_kernel): the GELU output that
        # feeds the first decoder block's second MLP matrix is pushed to ~1e5 by its bias -> an fp16 piece is inf -> a live
        # row's plane is not finite -> the kernel raises the model's flag.  The call that observes it fails loudly
        # (WB_ERR_STATE), the model switches to the exact-f32 skinny kernel, and the caller's retry decodes -- to the rows an
        # engine that never used the split kernel produces
        import os, subprocess, sys
        if os.environ.get("WHISPER_HIP_DECODER_SPLIT") == "0":
            w2 = dict(w)
            w2["decoder/block_0/mlp/mlp1/bias"] = np.asarray(w["decoder/block_0/mlp/mlp1/bias"], np.float32) + np.float32(1.0e5)
            e0 = wb.Whisper.from_tensors(w2)
            a = synth.synth_audio(16000 * 31, 33)
            _, wins0 = wb.waveform_to_tokens(e0, st, a, 16000, 6, 2)
            e0.close()
            print("ROWS " + repr(wins0))
        else:
            w2 = dict(w)
            w2["decoder/block_0/mlp/mlp1/bias"] = np.asarray(w["decoder/block_0/mlp/mlp1/bias"], np.float32) + np.float32(1.0e5)
            e2 = wb.Whisper.from_tensors(w2)
            a = synth.synth_audio(16000 * 31, 33)            # 3 windows x 6 beams = 18 live rows: batch mode, skinny GEMM
            failed = False
            try:
                wb.waveform_to_tokens(e2, st, a, 16000, 6, 2)
            except Exception as ex:                           # the binding raises on a negative status
                failed = "fp16" in str(ex) and "decode again" in str(ex)
                assert failed, str(ex)
            assert failed, "the range guard of the split-precision decoder GEMM did not trip"
            _, wins2 = wb.waveform_to_tokens(e2, st, a, 16000, 6, 2)      # the retry: exact-f32 decoder GEMMs
            e2.close()
            env = dict(os.environ); env["WHISPER_HIP_DECODER_SPLIT"] = "0"
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "dec_split_range"], env=env, capture_output=True, text=True)
            assert p.returncode == 0, p.stderr[-2000:]
            ref = eval([l for l in p.stdout.splitlines() if l.startswith("ROWS ")][-1][5:])
            assert wins2 == ref, (wins2, ref)
            assert len(wins2) == 3
    elif which in ("guard_chain", "guard_enc_deferred", "guard_session"):
        # tests/test_gpu_guard.py's checkpoints on the functional model: one MLP hidden unit pushed to 1e5 with an all-zero
        # row in the second matrix (exact result = the ordinary one; the fp16 pieces overflow: inf * 0 = NaN)
        def guarded(where):
            w2 = {k: np.array(v, copy=True) for k, v in synth.synth_weights(dims, seed=4242, eot_beta=0.0).items()}
            pfx = ("encoder" if where == "enc" else "decoder") + "/block_1/mlp"
            j = 4 * 128 - 1
            w2[pfx + "/mlp1/weight"][:, j] = 0.0
            w2[pfx + "/mlp1/bias"][j] = 1.0e5
            w2[pfx + "/mlp2/weight"][j, :] = 0.0
            return w2
        if which == "guard_enc_deferred":
            # wb_waveform_to_tokens defers the encoder guard to the decode's own synchronisation and decodes the batch again
            w2 = guarded("enc")
            e2, o2 = wb.Whisper.from_tensors(w2), OracleWhisper(w2)
            assert e2.encoder_gemm() == "f16x3"
            a = synth.synth_audio(16000 * 4, 77)
            ref, rw = otr.waveform_to_tokens(o2, pu.ost(st), a, 16000, 1, 6, return_windows=True)
            got, wins = wb.waveform_to_tokens(e2, st, a, 16000, 1, 6)
            assert e2.encoder_gemm() == "f32"
            assert got == ref and wins == rw, (got, ref)
            got, wins = wb.waveform_to_tokens(e2, st, a, 16000, 1, 6)       # (pooled session, guard word clear again)
            assert got == ref
            e2.close()
        else:
            w2 = guarded("dec")
            e2, o2 = wb.Whisper.from_tensors(w2), OracleWhisper(w2)
            assert e2.decoder_gemm() == "f16x3"
            n_win, wl = 18, 4000                                            # 18 live rows at d = 128: batch mode
            a = synth.synth_audio(wl * n_win, 78)
            starts = np.arange(n_win, dtype=np.int64) * wl
            lens = np.full(n_win, wl, dtype=np.int64)
            params = wb.decode_params(st, beam_size=1, max_depth=4)

            def decode():
                sess = wb.Session.begin(e2, a, starts, lens, max_beams=1)
                try:
                    sess.set_special_mask(st.is_special)
                    return sess.decode(params)
                finally:
                    sess.close()
            other = wb.Session.begin(e2, a, starts[:2], lens[:2], max_beams=1)      # bystander: 2 rows, never on the split kernel
            other.set_special_mask(st.is_special)
            failed = None
            try:
                decode()
            except Exception as ex:
                failed = ex
            assert failed is not None and getattr(failed, "status", 0) == -6 and "fp16" in str(failed), repr(failed)
            assert e2.decoder_gemm() == "f32"
            assert other.decode(params) is not None                         # not failed by the first session's trip
            other.close()
            rows = decode()                                                 # retry: exact-f32 decoder GEMMs
            for wi in (0, 17):
                mel = torch.from_numpy(wb.prep_audio(a[None, wi * wl:(wi + 1) * wl]))
                assert rows[wi] == otr.mels_to_tokens(o2, pu.ost(st), mel, 10, 1, 4), wi
            e2.close()
    elif which == "beam_batch":
        # batch mode with MORE than 32 live rows: 9 windows x 4 beams = 36 rows -> three 16-row tiles of the skinny
        # weight-stream GEMM (decode_batch.hip: v_mfma_f32_16x16x4_f32, split-K planes; with three or four tiles a thread
        # stores two (row, column quad) items of the block's plane), chunked cross-attention + combine (beams share a
        # window's cached K/V)
        a = synth.synth_audio(16000 * 100, 33)
        got, wins = wb.waveform_to_tokens(eng, st, a, 16000, 4, 5)
        ref, rw = otr.waveform_to_tokens(o, pu.ost(st), a, 16000, 4, 5, return_windows=True)
        assert len(rw) == 9, len(rw)
        assert wins == rw and got == ref, (wins, rw)
    elif which == "beam":
        a = synth.synth_audio(16000 * 2, 9)
        got, _ = wb.waveform_to_tokens(eng, st, a, 16000, 3, 6)
        ref = otr.waveform_to_tokens(o, pu.ost(st), a, 16000, 3, 6)
        assert got == ref, (got, ref)
    elif which == "beam16":
        # 9 - 16 live rows on the FUSED sublayer path (round 5): 3 windows x beam 5 = 15 rows -- the reference's live setting on
        # a 30 s chunk -- attention / cross-attention blocks per (head, row), the MLP block and the logits GEMV as two row
        # groups of 8 (grid.y / grid.z); WHISPER_HIP_FUSE16=0 sends the same rows through batch mode.  d = 128 and d = 384.
        a = synth.synth_audio(16000 * 31, 29)
        got, wins = wb.waveform_to_tokens(eng, st, a, 16000, 5, 6)
        ref, rw = otr.waveform_to_tokens(o, pu.ost(st), a, 16000, 5, 6, return_windows=True)
        assert len(rw) == 3, len(rw)
        assert wins == rw and got == ref, (wins, rw)
        dims = synth.micro_dims(n_state=384, n_head=6, n_layer=1, n_vocab=2053, n_audio_ctx=400)
        w2 = synth.synth_weights(dims, seed=61)
        e2, o2 = wb.Whisper.from_tensors(w2), OracleWhisper(w2)
        s2 = wb.SpecialTokens.for_vocab(2053)
        a2 = synth.synth_audio(16000 * 9, 47)            # n_audio_ctx = 400: windows of 3.9 s -> 9 windows; decode 3 of them
        p5 = wb.decode_params(s2, beam_size=5, max_depth=5)
        _, w2rows = wb.waveform_to_tokens(e2, s2, a2, 16000, params=p5, win_begin=2, win_end=5)
        _, r2rows = otr.waveform_to_tokens(o2, pu.ost(s2), a2, 16000, 5, 5, return_windows=True)
        assert w2rows == r2rows[2:5], (w2rows, r2rows[2:5])
        e2.close()
    elif which == "forward":
        a = synth.synth_audio(16000, 5)
        mel = np.concatenate([wb.prep_audio(a[None]), np.zeros((1, 80, 10), np.float32)], 2)
        toks = np.array([[st.start_of_transcript, st.language, st.transcribe, st.no_timestamps, 17, 300, 45]], np.int32)
        got = eng.forward(mel, toks)
        ref = o.forward(torch.from_numpy(mel), torch.from_numpy(toks)).numpy()
        assert np.abs(got - ref).max() < 1e-3, np.abs(got - ref).max()
    elif which == "geometry384":
        # the two-pass key ring of the fused cross-attention block (a window with more keys than one pass holds: 768 on
        # the device, 384 in this build): d = 384, doubled windows of 2 x 400 frames (C = 400 keys) + a shorter tail window,
        # through the persistent kernel (default) or the chain of one launch per sublayer (WHISPER_HIP_PERSIST=0)
        dims = synth.micro_dims(n_state=384, n_head=6, n_layer=2, n_vocab=2053, n_audio_ctx=400)
        w2 = synth.synth_weights(dims, seed=57)
        e2, o2 = wb.Whisper.from_tensors(w2), OracleWhisper(w2, frame_limit_x2=True)
        s2 = wb.SpecialTokens.for_vocab(2053)
        a = synth.synth_audio(16000 * 11, 43)
        e2.set_frame_limit(True)
        got, wins = wb.waveform_to_tokens(e2, s2, a, 16000, 1, 8)
        ref, rw = otr.waveform_to_tokens(o2, pu.ost(s2), a, 16000, 1, 8, return_windows=True)
        assert got == ref and wins == rw and len(wins) == 3, (got, ref, wins, rw)
        e2.close()
    elif which == "geometry":
        # wb_model_set_frame_limit(1): windows of 2 n_audio_ctx frames.  n_audio_ctx = 400 keeps the clip short while
        # both window lengths (3.9 s and 7.9 s) stay above the 3 s overlap (below it the reference's shift is 1 sample)
        dims = synth.micro_dims(n_state=128, n_head=2, n_layer=2, n_vocab=1031, n_audio_ctx=400)
        w2 = synth.synth_weights(dims, seed=11)
        e2, o2 = wb.Whisper.from_tensors(w2), OracleWhisper(w2, frame_limit_x2=True)
        a = synth.synth_audio(16000 * 6, 41)
        _, ref_limit_wins = wb.waveform_to_tokens(e2, st, a, 16000, 1, 6)
        assert ref_limit_wins == otr.waveform_to_tokens(OracleWhisper(w2), pu.ost(st), a, 16000, 1, 6, return_windows=True)[1]
        e2.set_frame_limit(True)
        got, wins = wb.waveform_to_tokens(e2, st, a, 16000, 1, 6)
        ref, rw = otr.waveform_to_tokens(o2, pu.ost(st), a, 16000, 1, 6, return_windows=True)
        assert got == ref and wins == rw and len(wins) < len(ref_limit_wins), (got, ref)
        e2.close()
    elif which.startswith("shape"):
        # the other decode-kernel template families: d = 384 (tiny.en's fused sublayer kernels), 512 (base.en's),
        # 768 (no fused path: per-matrix GEMVs, unfused cross-attention), one layer each, vocabulary not a tile multiple
        d = int(which[5:])
        dims = synth.micro_dims(n_state=d, n_head=d // 64, n_layer=1, n_vocab=2053, n_audio_ctx=400)
        w2 = synth.synth_weights(dims, seed=31 + d)
        e2, o2 = wb.Whisper.from_tensors(w2), OracleWhisper(w2)
        s2 = wb.SpecialTokens.for_vocab(2053)
        a = synth.synth_audio((16000 * 3 if d < 768 else 16000 * 3 // 2) + 123, 77)   # one window (emulated MFMA: d = 768 gets 1.5 s)
        for beam, depth in ((1, 7), (3, 4)):
            got, wins = wb.waveform_to_tokens(e2, s2, a, 16000, beam, depth)
            ref, rw = otr.waveform_to_tokens(o2, pu.ost(s2), a, 16000, beam, depth, return_windows=True)
            assert wins == rw and got == ref, (d, beam, wins, rw)
        e2.close()
    elif which == "bitwise":
        # a digest of raw output bytes: the summation orders are fixed (split-K planes folded in a fixed order, no float
        # atomics), so the digest may not depend on the order in which blocks / waves / lanes are scheduled
        import hashlib
        hsh = hashlib.sha256()
        a = synth.synth_audio(16000 * 21, 19)                             # two windows
        mel = np.concatenate([wb.prep_audio(a[None, :32000]), np.zeros((1, 80, 10), np.float32)], 2)
        toks = np.array([[st.start_of_transcript, st.language, st.transcribe, st.no_timestamps, 11, 503, 77, 9]], np.int32)
        hsh.update(np.ascontiguousarray(mel).tobytes())
        hsh.update(np.ascontiguousarray(eng.forward(mel, toks)).tobytes())
        starts, lens = wb.window_extents(len(a), 16000, wb.max_waveform_samples(1490))
        sess = wb.Session.begin(eng, a, starts, lens, max_beams=3)
        sess.set_special_mask(st.is_special)
        prompt = [st.start_of_transcript, st.language, st.transcribe, st.no_timestamps]
        for i, t in enumerate(prompt[:-1]):
            sess.step([t, t], [-1, -1] if i == 0 else [0, 1], [0, 1], apply_special_mask=False, k=0)
        ids, lps = sess.step([prompt[-1]] * 2, [0, 1], [0, 1], apply_special_mask=True, k=3)
        hsh.update(np.ascontiguousarray(ids).tobytes()); hsh.update(np.ascontiguousarray(lps).tobytes())
        # fork: three beams per window continue with their own top-3
        nt = [int(ids[w][j]) for w in (0, 1) for j in range(3)]
        ids2, lps2 = sess.step(nt, [0, 0, 0, 1, 1, 1], [0, 0, 0, 1, 1, 1], apply_special_mask=True, k=3)
        hsh.update(np.ascontiguousarray(lps2).tobytes())
        for slot in range(6):
            hsh.update(np.ascontiguousarray(sess.last_logprobs(slot)).tobytes())
        sess.close()
        got, _ = wb.waveform_to_tokens(eng, st, a, 16000, 1, 8)
        hsh.update(np.asarray(got, np.int32).tobytes())
        print("DIGEST", hsh.hexdigest())
    elif which == "sharded":
        import socket
        import torch.multiprocessing as mp
        a = synth.synth_audio(16000 * 52, 17)
        ref, rw = otr.waveform_to_tokens(o, pu.ost(st), a, 16000, 1, 8, return_windows=True)
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
        for p_ in procs:
            p_.start()
        res = [q.get(timeout=240) for _ in procs]
        for p_ in procs:
            p_.join(timeout=60)
            assert p_.exitcode == 0
        for rank, n_win, toks, wins in res:
            assert n_win == 5 and wins == rw and toks == ref, (rank, toks, ref)
    elif which == "prompted":
        # optional mode: the prompt conditioning the reference disabled (transcribe.rs:43-50, :188-199)
        from whisper_burn_amd import legacy
        a = synth.synth_audio(16000 * 28, 13)                            # 3 windows
        for beam in (1, 3):
            got, rows = legacy.waveform_to_tokens_prompted(eng, st, a, 16000, beam_size=beam, max_depth=7)
            ref, rrows = otr.waveform_to_tokens(o, pu.ost(st), a, 16000, beam, 7, return_windows=True,
                                                start_of_prev=st.start_of_prev)
            assert len(rows) == 3 and rows[0][0] == st.start_of_transcript and rows[1][0] == st.start_of_prev, rows
            assert rows == rrows and got == ref, (beam, rows, rrows)
    else:
        raise SystemExit(f"unknown check {which}")
    eng.close()
    print("EMU_CHECK_OK", which)


if __name__ == "__main__":
    main(sys.argv[1])
