/* whisper_hip.h -- C ABI of libwhisper_hip.so, the MI355X (gfx950) drop-in for the
 * tensor seams of Gadersd/whisper-burn.
 *
 * The reference has no FFI: its only seam is the `B: Backend` type parameter.  The
 * boundary is therefore cut at the three tensor call sites of src/transcribe.rs
 * (prep_audio :134, Whisper::forward_encoder :222, Whisper::forward_decoder :270) plus
 * the public Whisper::forward / waveform_to_text, and a Rust `extern "C"` shim keeps
 * the reference's signatures on top of these entry points.
 *
 * Conventions
 *  - every function returns WB_OK (0) or a negative wb_status; nothing aborts or
 *    throws across the boundary (the reference's assert!/panic sites are mapped to
 *    WB_ERR_SHAPE); wb_last_error() gives a thread-local message.
 *  - the caller owns every host buffer it passes; the library owns device memory
 *    behind opaque handles.  All pointers below are HOST pointers unless the
 *    parameter name ends in `_dev`.
 *  - all floating point data is IEEE f32 (the reference runs TchBackend<f32>,
 *    src/bin/transcribe/main.rs:80); token ids are int32.
 *  - a wb_model may be shared by threads: its weights never change after load.  The one
 *    piece of state it carries is the arithmetic switches (wb_model_encoder_gemm /
 *    wb_model_decoder_gemm), which can go from 1 to 0 once, atomically, when a range guard of
 *    the split-precision kernels trips (the guard words themselves are per session).  A
 *    wb_session is not thread-safe.
 *  - calls from different threads are safe and give the single-threaded results, but they TAKE
 *    TURNS on the GPU: every entry point that enqueues kernels holds its device's turn (one per
 *    GPU and process) from its first launch to its last synchronisation.  (On gfx950 a wave's packed-FP32
 *    instructions return wrong results while another kernel's f16 MFMAs run on the same
 *    SIMD: kernels of two calls must not share the device.  DESIGN.md section 9,
 *    tools/pk_mfma_probe.cpp.)  One process per GPU is the scaling model.
 */
#ifndef WHISPER_HIP_H
#define WHISPER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum wb_status {
  WB_OK = 0,
  WB_ERR_ARG = -1,     /* null pointer / bad enum / bad handle                     */
  WB_ERR_SHAPE = -2,   /* a shape contract the reference assert!s on was violated  */
  WB_ERR_IO = -3,      /* dump-dir / npy read error (load.rs:29-45 Box<dyn Error>) */
  WB_ERR_HIP = -4,     /* HIP runtime error (message carries hipGetErrorString)    */
  WB_ERR_OOM = -5,
  WB_ERR_STATE = -6    /* call sequence error on a session                         */
} wb_status;

/* compute_dtype for wb_model_load_*: the arithmetic the GEMMs run in. */
enum { WB_F32 = 0,     /* f32 results: exact-f32 MFMA, and the split-precision fp16 MFMA kernel (three products per pair,
                          f32 accumulate: f32-grade) for the encoder side -- see wb_model_encoder_gemm */
       WB_BF16 = 1 };  /* RETIRED in round 4 (wb_model_load_* return WB_ERR_ARG): plain bf16 inputs cannot hold the
                          path's 1e-3 logit tolerance; the 16-bit matrix path is the split-precision kernel above */

typedef struct wb_model wb_model;     /* Whisper<B>            src/model/mod.rs:41-45   */
typedef struct wb_session wb_session; /* per window-batch decode state (new: the
                                         reference has no KV cache, transcribe.rs:270) */

/* WhisperConfig = AudioEncoderConfig + TextDecoderConfig, src/model/mod.rs:16-20,
 * :73-80, :164-171 */
typedef struct wb_dims {
  int32_t n_mels, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
  int32_t n_vocab, n_text_ctx, n_text_state, n_text_head, n_text_layer;
} wb_dims;

/* ---- model ---------------------------------------------------------------------- */

/* load_whisper(path), src/model/load.rs:295-310: reads the dump directory written by
 * python/dump.py (1-D f32 .npy files, shape stored as leading floats). */
int wb_model_load_dump_dir(const char* dir, int device, int compute_dtype, wb_model** out);

/* Same model from tensors already in host memory, named by their dump-dir relative
 * path without ".npy" (e.g. "encoder/block_0/attn/query/weight"); lets a Rust caller
 * hand over the tensors of a Burn record (src/bin/transcribe/main.rs:63-70).
 * shapes[i] points at ranks[i] dims.  Scalars (n_head, n_layer, eps ...) have rank 1. */
int wb_model_load_tensors(const char* const* names, const float* const* data,
                          const int64_t* const* shapes, const int32_t* ranks, int n,
                          int device, int compute_dtype, wb_model** out);

/* load_whisper_model_file, src/bin/transcribe/main.rs:63-70 (+ WhisperConfig::load of `<name>.cfg`, :116-123):
 * the converter's output (src/bin/convert/main.rs:17-19, :51), NamedMpkGzFileRecorder<FullPrecisionSettings> =
 * gzip(MessagePack with field names) of the Whisper module record.  cfg_path may be NULL (head counts then
 * follow from n_state / 64).  Burn 0.9.0 is not vendored with the reference: the reader walks the record
 * structurally (any map with "value" + "shape" is a tensor named by its key path) -- format parity unpinned. */
int wb_model_load_burn_record(const char* mpk_gz_path, const char* cfg_path, int device, int compute_dtype,
                              wb_model** out);
/* The same reader without a device: calls fn(user, dump-style name, data, shape, rank) per tensor (host logic,
 * testable without a GPU); a non-zero return of fn stops the walk and is returned. */
typedef int (*wb_tensor_fn)(void* user, const char* name, const float* data, const int64_t* shape, int32_t rank);
int wb_burn_record_read(const char* mpk_gz_path, const char* cfg_path, wb_tensor_fn fn, void* user);

/* Whisper::encoder_ctx_size / decoder_ctx_size (mod.rs:64-70) and the rest of the config. */
int wb_model_dims(const wb_model* m, wb_dims* out);
void wb_model_free(wb_model* m);

/* LayerNorm epsilon placement: 0 = (x-mu)/(sqrt(var)+eps)  [Burn 0.9.0 @ fb2a71bb, default]
 *                              1 = (x-mu)/sqrt(var+eps)    [later Burn releases, HF]      */
int wb_model_set_ln_variant(wb_model* m, int eps_inside_sqrt);

/* Mel frames per window: 0 = at most n_audio_ctx FRAMES, exactly as the reference asserts (mod.rs:236-241; its
 *                            windows are therefore 14.9 s, transcribe.rs:32-34)                     [default]
 *                        1 = at most n_audio_ctx encoder POSITIONS = 2 n_audio_ctx frames: Whisper's own 30 s
 *                            window ("perf geometry": T = 3000, C = 1500).  NOT reference behaviour -- the
 *                            reference panics there; window length, clipping and padding follow the same
 *                            formulas (transcribe.rs:32-34, :171-177) with the larger bound. */
int wb_model_set_frame_limit(wb_model* m, int whisper_geometry);

/* Arithmetic of the encoder-side Linear layers of this model (chosen at load time; 1 can turn into 0 ONCE, see below):
 *   0 = exact-f32 MFMA (v_mfma_f32_32x32x2_f32)
 *   1 = split precision: three fp16 MFMAs per product on fp16 hi / lo pieces, f32 accumulation -- f32-grade results
 *       (default for f32 models; WHISPER_HIP_ENCODER_SPLIT=0 at load time selects 0, and a model whose activations leave
 *       fp16's range, |x| >= 65504, falls back to 0 by itself: the pass is repeated, the answer changes from then on)
 * Replaces nothing in the reference (its Linear is Burn's, mod.rs:377-379); a caller reports it next to its timings. */
int wb_model_encoder_gemm(const wb_model* m);

/* Arithmetic of the decoder's Linear layers in BATCH MODE (more than 8 -- at d <= 512: 16 -- live rows per step):
 *   0 = exact-f32 MFMA (v_mfma_f32_16x16x4_f32)
 *   1 = split precision on fp16 hi / lo weight tiles (v_mfma_f32_16x16x32_f16 x 3, f32 accumulation; default for f32 models;
 *       WHISPER_HIP_DECODER_SPLIT=0 at load time selects 0).  1 can turn into 0 ONCE: when a decoder activation leaves fp16's
 *       range the decode call that observes it fails with WB_ERR_STATE (its rows are invalid), the model switches to 0 and
 *       the caller's retry succeeds.  The guard word is per SESSION: only the session whose rows were invalid fails.
 * Replaces nothing in the reference (mod.rs:345-350 runs Burn's Linear). */
int wb_model_decoder_gemm(const wb_model* m);

/* ---- stateless, reference-shaped entry points (the parity surface) -------------- */

/* max_waveform_samples(n_frame_max), src/audio.rs:12-17 */
int64_t wb_max_waveform_samples(int64_t n_frame_max);

/* prep_audio(waveform [1,n], sample_rate) -> [1,80,n/160], src/audio.rs:34-56.
 * mel must hold 80*(n/160) floats; *n_frames receives n/160.  n < 400 -> WB_ERR_SHAPE
 * (audio.rs:292 assert). */
int wb_prep_audio(int device, const float* pcm, int64_t n, double sample_rate, float* mel,
                  int64_t* n_frames);

/* ---- WAV ingest with the reference's sample scaling (next to the hot path) ---------- */

/* load_audio_waveform, src/bin/transcribe/main.rs:31-55 (hound): header of a RIFF/WAVE file.
 * n_samples is per channel.  Any out pointer may be NULL. */
int wb_wav_info(const char* path, int64_t* n_samples, int32_t* sample_rate, int32_t* channels,
                int32_t* bits, int32_t* is_float);
/* The samples as f32: integer PCM (8 / 16 / 24 / 32 bit) as s / (2^(bits-1) - 1) (main.rs:45-52), IEEE
 * float as stored (main.rs:48).  A sample rate other than 16 000 Hz or more than one channel ->
 * WB_ERR_SHAPE (the asserts of main.rs:42-43); capacity < samples -> WB_ERR_ARG. */
int wb_wav_read_f32(const char* path, float* out, int64_t capacity, int64_t* n_samples);
/* wb_wav_read_f32 without the 16 kHz assert (still mono): the input of wb_resample_dev. */
int wb_wav_read_f32_any_rate(const char* path, float* out, int64_t capacity, int64_t* n_samples);

/* Sample-rate conversion in HBM for inputs the reference's CLI rejects (main.rs:42; its README sends them through
 * `sox`, README.md:69-74 -- the bundled audio.wav is 22 050 Hz).  Rational polyphase resampler with the design of
 * SciPy's resample_poly(x, up, down): up/down = rate_out/rate_in reduced, Kaiser(beta 5) windowed sinc of half
 * length 10*max(up,down), cut-off 1/max(up,down) of Nyquist, zero extension; taps designed in f64, f32 arithmetic.
 *   wb_resample_len     ceil(n_in*up/down), or -1 on a bad argument / a reduced ratio above 1600
 *   wb_resample_filter  the taps (host only; taps may be NULL to query n_taps = 20*max(up,down)+1, up, down)
 *   wb_resample_dev     src_dev[n_in] -> dst_dev[*n_out] (DEVICE pointers), one pass, synchronises before returning */
int64_t wb_resample_len(int64_t n_in, int32_t rate_in, int32_t rate_out);
int wb_resample_filter(int32_t rate_in, int32_t rate_out, float* taps, int32_t capacity, int32_t* n_taps, int32_t* up,
                       int32_t* down);
int wb_resample_dev(int device, const float* src_dev, int64_t n_in, int32_t rate_in, int32_t rate_out, float* dst_dev,
                    int64_t capacity, int64_t* n_out);

/* The same scaling for 16-bit PCM already resident in HBM: dst_dev[i] = src_dev[i] / 32767 (correctly
 * rounded, bit-identical to the host path); halves the host->device bytes of wb_waveform_to_tokens_dev. */
int wb_pcm_s16_to_f32_dev(int device, const int16_t* src_dev, int64_t n, float* dst_dev);

/* The frontend alone, batched and device-resident: the window iterator of waveform_to_mel_tensor
 * (src/transcribe.rs:114-138: window w = pcm[starts[w], +lens[w]) -> prep_audio) followed by the clip /
 * zero-pad of mels_to_text (src/transcribe.rs:171-177: keep clip_frames frames, append `padding` zero
 * frames), for all windows in one launch pair.  pcm_dev / mel_dev are DEVICE pointers; window w is written
 * to mel_dev + w * win_stride as [80][row_stride] (row_stride % 4 == 0, >= frames_out[w]); frames_out[w]
 * = min(lens[w] / 160, clip_frames) + padding.  iters >= 1 repeats the pass; elapsed_ms (optional)
 * receives the HIP-event time of all passes on the launch stream (the mel-frames/s measurement). */
int wb_waveform_to_mels_dev(int device, const float* pcm_dev, int64_t n_samples, double sample_rate,
                            const int64_t* starts, const int64_t* lens, int32_t n_windows, int32_t clip_frames,
                            int32_t padding, float* mel_dev, int64_t win_stride, int32_t row_stride,
                            int32_t* frames_out, int32_t iters, double* elapsed_ms);

/* Whisper::forward_encoder(mel [B,80,T]) -> [B,C,d], C=(T-1)/2+1, src/model/mod.rs:52-54,
 * :228-260.  T > n_audio_ctx -> WB_ERR_SHAPE (mod.rs:236-241). */
int wb_forward_encoder(wb_model* m, const float* mel, int B, int T, float* out);

/* Whisper::forward_decoder(tokens [n,L], encoder_output [n,C,d]) -> logits [n,L,V],
 * src/model/mod.rs:56-62, :131-157.  Stateless: cross-attention K/V are re-projected
 * (mod.rs:482-490).  L > n_text_ctx -> WB_ERR_SHAPE (mod.rs:134-139). */
int wb_forward_decoder(wb_model* m, const int32_t* tokens, int n, int L, const float* enc, int C,
                       float* logits);

/* Whisper::forward(mel [B,80,T], tokens [B,L]) -> logits [B,L,V], src/model/mod.rs:48-50 */
int wb_forward(wb_model* m, const float* mel, int B, int T, const int32_t* tokens, int L,
               float* logits);

/* ---- stateful fast path (result-equivalent to transcribe.rs:148-312) ------------ */

/* Decode constants the reference hard-codes; wb_decode_params_default() fills them in. */
typedef struct wb_decode_params {
  int32_t beam_size;            /* transcribe.rs:232 (5).  1 == greedy */
  int32_t max_depth;            /* transcribe.rs:233 (100)                            */
  int32_t padding;              /* transcribe.rs:33  (10 zero mel frames)             */
  int32_t overlap_seconds;      /* transcribe.rs:120 (3)                              */
  int32_t max_n_offsets;        /* transcribe.rs:57  (40)                             */
  int32_t min_n_overlaps;       /* transcribe.rs:57  (3)                              */
  int32_t mask_until_len;       /* transcribe.rs:271 (5): special mask while len <= 5 */
  int32_t max_batch_windows;    /* engine knob: windows encoded/decoded together (0 = all) */
  /* special tokens by id (transcribe.rs:179-185; looked up by name in tokenizer.json
   * by the reference -- the tokenizer stays on the caller's side of the boundary) */
  int32_t tok_start_of_transcript, tok_language, tok_transcribe, tok_no_timestamps,
      tok_end_of_text;
} wb_decode_params;
void wb_decode_params_default(wb_decode_params* p);

/* mel (+clip, +`padding` zero frames) -> encoder -> cross-attention K/V once, for
 * n_windows windows cut from `pcm` at [starts[i], starts[i]+lens[i]).
 * transcribe.rs:134, :171-177, :222.  max_beams = live beams per window. */
int wb_session_begin(wb_model* m, const float* pcm, int64_t n_pcm, const int64_t* starts,
                     const int64_t* lens, int n_windows, int max_beams, int padding,
                     wb_session** out);
/* Same, from already-prepared mel windows mel[i] = [80, T[i]] packed back to back. */
int wb_session_begin_mel(wb_model* m, const float* mel, const int32_t* T, int n_windows,
                         int max_beams, int padding, wb_session** out);

/* is_special[V] != 0 where bpe.is_special(id); transcribe.rs:243-251 */
int wb_session_set_special_mask(wb_session* s, const uint8_t* is_special);

/* One decode step for n live beams (KV-cached equivalent of the closure
 * `beamsearch_next`, transcribe.rs:253-307).  Beam i of this step continues the beam
 * that occupied slot parent[i] in the previous step (-1: a fresh, empty beam) of window
 * window[i], and appends new_tokens[i].  If k > 0 the k best continuations of each beam
 * by (log-prob descending, token id ascending) are returned: log_softmax over the
 * (optionally special-masked, transcribe.rs:271-275) logits of the last position.
 * WB_ERR_STATE with "... left fp16's range ..." in wb_last_error(): with more than 8 live rows (16 at d <= 512) the decoder's
 * Linear layers run on fp16 hi / lo weight pieces (f32-grade results); an activation of |x| >= 65504 there makes this
 * step's rows invalid.  The call says so, the model switches to the exact-f32 kernels for good, and decoding again
 * (a new session, or wb_waveform_to_tokens again) succeeds.  Whisper's decoder activations are O(10): a backstop. */
int wb_session_step(wb_session* s, const int32_t* new_tokens, const int32_t* parent,
                    const int32_t* window, int n, int apply_special_mask, int k, int32_t* top_ids,
                    float* top_logprobs);

/* Debug/parity: full log-prob row [V] of beam slot i after the last step. */
int wb_session_last_logprobs(wb_session* s, int slot, float* out);
/* Debug/parity: encoder output of window w, [C_w, d]; *C receives C_w. */
int wb_session_encoder_output(wb_session* s, int w, float* out, int32_t* C);
void wb_session_free(wb_session* s);

/* mels_to_text (transcribe.rs:148-383) without the tokenizer, for a batch of windows:
 * beam search (src/beam.rs:9-110) driven by wb_session_step.  out_tokens holds
 * n_windows rows of `row_stride` ints; out_lens[i] = sequence length of window i
 * (prompt included, transcribe.rs:309-312). */
int wb_session_decode(wb_session* s, const wb_decode_params* p, int32_t* out_tokens,
                      int32_t row_stride, int32_t* out_lens);

/* Optional mode (not the reference's live behaviour): the same beam search from a caller-supplied initial
 * sequence instead of the four-token prompt of transcribe.rs:203 -- the building block of the prompt
 * conditioning the reference wrote and then disabled (transcribe.rs:188-199, shadowed at :201).  The special-token
 * mask still applies while the sequence length is <= mask_until_len (transcribe.rs:271-275), i.e. never for a
 * prompt longer than that.  row_stride >= prompt_len + max_depth. */
int wb_session_decode_prompt(wb_session* s, const wb_decode_params* p, const int32_t* prompt, int32_t prompt_len,
                             int32_t* out_tokens, int32_t row_stride, int32_t* out_lens);

/* Optional mode: waveform_to_text with that prompt conditioning switched back on -- every window after the first
 * starts from [tok_start_of_prev, the last n_prev_tokens (reference: 5) non-special tokens of the transcript so far,
 * start_of_transcript, language, transcribe, no_timestamps] (transcribe.rs:43-50, :188-199, :203), and its row
 * (prompt included, as mels_to_text returns it) is stitched with find_chunk_overlap (:56-63).  Windows depend on
 * their predecessors, so they are decoded one at a time; win_tokens holds one row of `row_stride` ints per window,
 * row_stride >= 1 + n_prev_tokens + 4 + max_depth.  PCM on the host. */
int wb_waveform_to_tokens_prompted(wb_model* m, const float* pcm, int64_t n, int sample_rate,
                                   const wb_decode_params* p, const uint8_t* is_special, int32_t tok_start_of_prev,
                                   int32_t n_prev_tokens, int32_t* win_tokens, int32_t row_stride, int32_t* win_lens,
                                   int32_t* stitched, int64_t stitched_cap, int64_t* n_stitched);

/* The same beam search (src/beam.rs:9-110 driving the closure of transcribe.rs:253-307) over a
 * caller-supplied step function with wb_session_step's contract -- the host logic without the
 * GPU, e.g. for a Rust caller that owns its own model, and for CPU tests of the bookkeeping. */
typedef int (*wb_step_fn)(void* user, const int32_t* new_tokens, const int32_t* parent,
                          const int32_t* window, int n, int apply_special_mask, int k,
                          int32_t* top_ids, float* top_logprobs);
int wb_beam_search(const wb_decode_params* p, int n_windows, int n_vocab, wb_step_fn step, void* user,
                   int32_t* out_tokens, int32_t row_stride, int32_t* out_lens);
/* The same search with the bookkeeping of src/beam.rs:39-79 done by the DEVICE kernel that wb_session_decode chains between
 * decode steps (beam_size > 1; WHISPER_HIP_BEAM_CHAIN=0 keeps the host loop above), driven by a caller's step function: a test
 * hook that lets the device restatement be compared with the host one on scripted log-prob rows (exact ties, finished beams).
 * n_windows <= 64; `device` only hosts three small buffers (no model). */
int wb_beam_search_device(int device, const wb_decode_params* p, int n_windows, int n_vocab, wb_step_fn step,
                          void* user, int32_t* out_tokens, int32_t row_stride, int32_t* out_lens);

/* waveform_to_text (transcribe.rs:23-74) without the tokenizer: windows
 * (transcribe.rs:114-128), per-window decode, token-overlap stitch
 * (find_chunk_overlap, transcribe.rs:76-110).  Only windows [win_begin, win_end) are
 * decoded (multi-GPU sharding: SURVEY 8e); pass 0, -1 for all.  Per-window token rows go
 * to win_tokens [n_local, row_stride] / win_lens; when stitched != NULL the stitched
 * stream of the local windows is written there (capacity stitched_cap, length *n_stitched). */
int wb_waveform_to_tokens(wb_model* m, const float* pcm, int64_t n, int sample_rate,
                          const wb_decode_params* p, const uint8_t* is_special, int win_begin,
                          int win_end, int32_t* win_tokens, int32_t row_stride, int32_t* win_lens,
                          int32_t* stitched, int64_t stitched_cap, int64_t* n_stitched);

/* Same with the waveform already resident in device memory (`pcm_dev` is a DEVICE pointer
 * on the model's GPU; no copy is made -- the mel kernel reads the windows in place). */
int wb_waveform_to_tokens_dev(wb_model* m, const float* pcm_dev, int64_t n, int sample_rate,
                              const wb_decode_params* p, const uint8_t* is_special, int win_begin,
                              int win_end, int32_t* win_tokens, int32_t row_stride, int32_t* win_lens,
                              int32_t* stitched, int64_t stitched_cap, int64_t* n_stitched);

/* ---- multi-GPU: windows sharded over ranks -------------------------------------------------------
 * The reference is single-device and decodes its windows one after another (transcribe.rs:35-66); they are independent
 * (the previous-window prompt is discarded, transcribe.rs:195-201), so rank r of R decodes the contiguous block
 * [ceil(r K / R), ceil((r + 1) K / R)) of the K windows and the ranks exchange ONE fixed-shape buffer of token rows;
 * every rank then folds the stitch (transcribe.rs:56-63) over all K rows -- identical to world size 1 by construction.
 *
 * The exchange is an all-gather the caller supplies: `send` holds bytes_per_rank bytes of this rank, `recv` receives
 * world x bytes_per_rank bytes in rank order (host memory both).  wb_comm_allgather is the built-in RCCL transport
 * (RCCL over xGMI; librccl is opened at run time, on first use): rank 0 makes an id with wb_comm_unique_id and ships its
 * 128 bytes to the other ranks by its own means (a file, a socket, MPI, torch.distributed ...), every rank calls
 * wb_comm_init with its own device, and passes (wb_comm_allgather, comm) below. */
typedef int (*wb_allgather_fn)(void* user, const void* send, void* recv, int64_t bytes_per_rank);
typedef struct wb_comm wb_comm;
int wb_comm_unique_id(uint8_t* id128);
int wb_comm_init(const uint8_t* id128, int rank, int world, int device, wb_comm** out);
void wb_comm_free(wb_comm* c);
int wb_comm_allgather(void* comm, const void* send, void* recv, int64_t bytes_per_rank);

/* The block of rank `rank`: [*lo, *hi) = [ceil(rank K / world), ceil((rank + 1) K / world)). */
int wb_shard_partition(int64_t n_windows, int rank, int world, int64_t* lo, int64_t* hi);

/* waveform_to_text (transcribe.rs:23-74) without the tokenizer, sharded: this rank decodes its block of windows of the
 * WHOLE waveform `pcm` (host memory, or device memory of the model's GPU when pcm_on_device != 0), all-gathers the
 * rows and stitches.  On return EVERY rank holds all K per-window rows (win_tokens [K][row_stride], win_lens [K],
 * K <= win_cap) and the stitched stream.  world == 1 needs no all-gather (allgather may be NULL).
 * Failure: a rank whose local decode fails (out of memory, a HIP error) STILL enters the all-gather -- its rows carry a
 * sentinel -- so no rank is left blocked in the collective; every rank then returns an error (the failing rank its own
 * status, the others WB_ERR_STATE naming the rank).  Argument errors are reported before the collective on the rank that
 * has them: validate identically on every rank. */
int wb_waveform_to_tokens_sharded(wb_model* m, const float* pcm, int pcm_on_device, int64_t n, int sample_rate,
                                  const wb_decode_params* p, const uint8_t* is_special, int rank, int world,
                                  wb_allgather_fn allgather, void* user, int32_t* win_tokens, int32_t row_stride,
                                  int32_t* win_lens, int64_t win_cap, int32_t* stitched, int64_t stitched_cap,
                                  int64_t* n_stitched);

/* Window extents of waveform_to_mel_tensor (transcribe.rs:114-128).  Returns the window
 * count; fills starts/lens when non-NULL (capacity cap). */
int64_t wb_window_extents(int64_t n_samples, int sample_rate, int64_t window_len, int overlap_seconds,
                          int64_t* starts, int64_t* lens, int64_t cap);

/* find_chunk_overlap(prev, curr, max_n_offsets, min_n_overlaps), transcribe.rs:76-110.
 * Returns 1 and sets the indices if an overlap >= min_n_overlaps was found, else 0. */
int wb_find_chunk_overlap(const int32_t* prev, int64_t n_prev, const int32_t* curr, int64_t n_curr,
                          int max_n_offsets, int min_n_overlaps, int64_t* prev_index,
                          int64_t* curr_index);

/* Fold the stitch (transcribe.rs:56-63) over per-window token rows in window order. */
int wb_stitch_windows(const int32_t* win_tokens, int32_t row_stride, const int32_t* win_lens,
                      int n_windows, int max_n_offsets, int min_n_overlaps, int32_t* out,
                      int64_t cap, int64_t* n_out);

/* The reference's retired greedy decoder kept its repetition detectors (transcribe.rs:385-447, dead code there):
 *   wb_first_repetition_end        :385-393   (period > n, a usize underflow panic there -> WB_ERR_ARG)
 *   wb_repetition_period           :395-417   returns the period, 0 for None
 *   wb_find_repeated_tokens_index  :419-447   returns 1 and the (first, second) window indices, else 0
 * They serve the optional legacy greedy mode (whisper_burn_amd.legacy.legacy_greedy over the session API). */
int64_t wb_first_repetition_end(const int32_t* tokens, int64_t n, int64_t period);
int64_t wb_repetition_period(const int32_t* tokens, int64_t n, int64_t min_repetitions);
int wb_find_repeated_tokens_index(const int32_t* tokens, int64_t n, int64_t window_size, int64_t min_repeat_count,
                                  int64_t* first_repeat_index, int64_t* end);

/* The constant tables of the frontend, built on the host in the reference's f32 op order: periodic Hann window
 * (hann_window_device, audio.rs:272-278) and the dense [80][201] Slaney filterbank (get_mel_filters_device,
 * audio.rs:67-143; the kernel keeps it sparse).  No GPU needed; for table-vs-oracle tests. */
int wb_mel_constants(double sample_rate, float* hann400, float* filters_80x201);

/* ---- measurement hooks ----------------------------------------------------------- */

/* Per-stage device time (ms, HIP events on the engine's stream) accumulated since the
 * last reset: [0]=mel [1]=encoder [2]=cross-KV [3]=decode steps [4]=number of decode
 * steps [5]=mel kernel launches [6]=logits kernel ms [7]=logits kernel launches. */
int wb_profile_enable(int on);
int wb_profile_read(double* out8, int reset);

/* Per-kernel statistics of the decode-step launches made while profiling was on: every launch carries its own
 * start / stop HIP events (the dispatch's begin -> end, what `rocprofv3 --kernel-trace` reports) and the
 * algorithmic bytes it streams (weights + cached K/V).  Fills at most `cap` entries, returns the number of
 * kernel classes that ran (may exceed cap).  New: the reference has no profiling hooks. */
typedef struct wb_kernel_stat {
  char name[96];
  int64_t calls;
  double total_ms;      /* sum of the launches' own durations */
  double algo_bytes;    /* sum of their algorithmic bytes     */
} wb_kernel_stat;
int wb_profile_kernels(wb_kernel_stat* out, int cap, int reset);

const char* wb_last_error(void);
const char* wb_version(void);

#ifdef __cplusplus
}
#endif
#endif /* WHISPER_HIP_H */
