// C ABI of libwhisper_hip.so: model handles and the stateless, reference-shaped entry points.
// (The stateful session / decode driver entry points live in session.cpp / transcribe.cpp.)
#include <cstring>
#include <mutex>

#include "engine.h"
#include "session.h"

namespace wb {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}
const char* get_error() { return g_err.c_str(); }

int DevMem::alloc(size_t n) {
  release();
  if (n == 0) n = 256;
  hipError_t e = hipMalloc(&p, n);
  if (e != hipSuccess) {
    p = nullptr;
    set_error("hipMalloc(%zu) failed: %s", n, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? WB_ERR_OOM : WB_ERR_HIP;
  }
  bytes = n;
  return WB_OK;
}
int DevMem::ensure(size_t n) {
  if (n <= bytes && p) return WB_OK;
  return alloc(n + n / 8);
}
int DevMem::ensure_zeroed(size_t n, hipStream_t st) {
  if (n <= bytes && p) return WB_OK;
  WB_TRY(alloc(n + n / 8));
  // on the OWNER's stream (a fill on the legacy stream fails while another session of the process is capturing a step graph:
  // "would make the legacy stream depend on a capturing ... stream" -- tests/test_gpu_concurrency.py); wait here, once per
  // (re)allocation, so that no later cache append can be overtaken by the fill
  WB_HIP(hipMemsetAsync(p, 0, bytes, st));
  WB_HIP(hipStreamSynchronize(st));
  return WB_OK;
}
void DevMem::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  bytes = 0;
}

// One lock per process for the stateless entry points (they share the model's scratch workspace).
static std::mutex g_stateless_mu;

}  // namespace wb

using namespace wb;

extern "C" {

const char* wb_last_error(void) { return wb::get_error(); }
#if defined(HIPEMU) && defined(HIPEMU_PROD_GEOMETRY)
const char* wb_version(void) { return "whisper_hip 0.1 (hipemu functional model, prod geometry -- NOT a product build)"; }
#elif defined(HIPEMU)   // tools/hipemu: the host build against the functional model (test infrastructure, refused by the binding)
const char* wb_version(void) { return "whisper_hip 0.1 (hipemu functional model -- NOT a product build)"; }
#else
const char* wb_version(void) { return "whisper_hip 0.1 (gfx950)"; }
#endif

int wb_model_load_dump_dir(const char* dir, int device, int compute_dtype, wb_model** out) {
  WB_REQUIRE(dir && out, WB_ERR_ARG, "wb_model_load_dump_dir: null argument");
  TensorMap tm;
  WB_TRY(read_dump_dir(dir, &tm));
  return build_model(tm, device, compute_dtype, out);
}

int wb_model_load_tensors(const char* const* names, const float* const* data, const int64_t* const* shapes,
                          const int32_t* ranks, int n, int device, int compute_dtype, wb_model** out) {
  WB_REQUIRE(names && data && shapes && ranks && out && n > 0, WB_ERR_ARG, "wb_model_load_tensors: null argument");
  TensorMap tm;
  for (int i = 0; i < n; i++) {
    WB_REQUIRE(names[i] && data[i] && shapes[i] && ranks[i] >= 1, WB_ERR_ARG, "tensor %d: null / bad rank", i);
    HostTensor t;
    t.shape.assign(shapes[i], shapes[i] + ranks[i]);
    t.data = data[i];
    tm[names[i]] = std::move(t);
  }
  return build_model(tm, device, compute_dtype, out);
}

int wb_model_dims(const wb_model* m, wb_dims* out) {
  WB_REQUIRE(m && out, WB_ERR_ARG, "wb_model_dims: null argument");
  *out = m->dims;
  return WB_OK;
}

int wb_model_set_ln_variant(wb_model* m, int eps_inside_sqrt) {
  WB_REQUIRE(m, WB_ERR_ARG, "wb_model_set_ln_variant: null model");
  m->ln_eps_inside_sqrt = eps_inside_sqrt ? 1 : 0;
  return WB_OK;
}

int wb_model_set_frame_limit(wb_model* m, int whisper_geometry) {
  WB_REQUIRE(m, WB_ERR_ARG, "wb_model_set_frame_limit: null model");
  m->frame_limit_x2 = whisper_geometry ? 1 : 0;
  return WB_OK;
}

int wb_model_encoder_gemm(const wb_model* m) {
  if (!m) return WB_ERR_ARG;
  return m->split_active() ? 1 : 0;        // (0 as well once the range guard of the split kernel has tripped)
}

int wb_model_decoder_gemm(const wb_model* m) {
  if (!m) return WB_ERR_ARG;
  return m->dec_split_active() ? 1 : 0;    // (0 as well once a session's decoder range guard has tripped)
}

void wb_model_free(wb_model* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  session_pool_purge(m);
  if (m->stream) (void)hipStreamDestroy(m->stream);
  delete m;
}

int wb_mel_constants(double sample_rate, float* hann400, float* filters_80x201) {
  // host only: the tables the mel kernel uses (hann_window_device audio.rs:272-278, get_mel_filters_device :67-143)
  WB_REQUIRE(hann400 && filters_80x201, WB_ERR_ARG, "wb_mel_constants: null argument");
  auto t = std::make_unique<MelTables>();
  WB_REQUIRE(mel_tables_build(sample_rate, t.get()) == 0, WB_ERR_SHAPE,
             "mel filterbank for sample_rate %g has a row wider than %d taps", sample_rate, MEL_MAX_TAPS);
  memcpy(hann400, t->hann, sizeof(t->hann));
  memset(filters_80x201, 0, sizeof(float) * MEL_N_MELS * MEL_N_BINS);
  for (int m = 0; m < MEL_N_MELS; m++)
    for (int k = 0; k < t->tap_len[m]; k++) filters_80x201[m * MEL_N_BINS + t->tap_start[m] + k] = t->tap_w[m * MEL_MAX_TAPS + k];
  return WB_OK;
}

int64_t wb_max_waveform_samples(int64_t n_frame_max) {
  // audio.rs:12-17 with N_FFT = 400 (even)
  return MEL_HOP * (n_frame_max + 1) + (MEL_N_FFT % 2) - 1;
}

int wb_prep_audio(int device, const float* pcm, int64_t n, double sample_rate, float* mel, int64_t* n_frames) {
  WB_REQUIRE(pcm && mel, WB_ERR_ARG, "wb_prep_audio: null argument");
  WB_REQUIRE(n >= MEL_N_FFT, WB_ERR_SHAPE, "prep_audio: %lld samples < n_fft = 400 (audio.rs:292)", (long long)n);
  WB_REQUIRE(n < ((int64_t)1 << 31), WB_ERR_SHAPE, "prep_audio: window too long");
  wb::GpuTurn turn(device);
  WB_HIP(hipSetDevice(device));
  const MelTables* tabs;
  WB_TRY(get_mel_tables(device, sample_rate, &tabs));
  const int T = (int)(n / MEL_HOP);
  DevMem d_pcm, d_out, d_win, d_max;
  WB_TRY(d_pcm.alloc((size_t)n * 4));
  WB_TRY(d_out.alloc((size_t)80 * T * 4));
  WB_TRY(d_win.alloc(sizeof(MelWindow)));
  WB_TRY(d_max.alloc((size_t)mel_bmax_stride(T) * 2 * 4));
  MelWindow w{0, (int32_t)n, T, T, 0};
  hipStream_t st = nullptr;
  WB_HIP(hipMemcpyAsync(d_pcm.p, pcm, (size_t)n * 4, hipMemcpyHostToDevice, st));
  WB_HIP(hipMemcpyAsync(d_win.p, &w, sizeof(w), hipMemcpyHostToDevice, st));
  launch_mel_spectrogram(st, d_pcm.as<float>(), d_win.as<MelWindow>(), 1, T, tabs, d_out.as<float>(),
                         (int64_t)80 * T, T, d_max.as<float>(), 0, T);
  launch_mel_finalize(st, d_win.as<MelWindow>(), 1, d_out.as<float>(), (int64_t)80 * T, T, d_max.as<float>(), T);
  WB_HIP(hipGetLastError());
  WB_HIP(hipMemcpyAsync(mel, d_out.p, (size_t)80 * T * 4, hipMemcpyDeviceToHost, st));
  WB_HIP(hipStreamSynchronize(st));
  if (n_frames) *n_frames = T;
  return WB_OK;
}

int wb_waveform_to_mels_dev(int device, const float* pcm_dev, int64_t n_samples, double sample_rate,
                            const int64_t* starts, const int64_t* lens, int32_t n_windows, int32_t clip_frames,
                            int32_t padding, float* mel_dev, int64_t win_stride, int32_t row_stride,
                            int32_t* frames_out, int32_t iters, double* elapsed_ms) {
  WB_REQUIRE(pcm_dev && starts && lens && mel_dev && n_windows > 0, WB_ERR_ARG, "wb_waveform_to_mels_dev: bad argument");
  WB_REQUIRE(clip_frames > 0 && padding >= 0 && iters >= 1, WB_ERR_ARG, "wb_waveform_to_mels_dev: bad clip/padding/iters");
  std::vector<MelWindow> wins(n_windows);
  int maxF = 0, maxT = 0;
  for (int w = 0; w < n_windows; w++) {
    WB_REQUIRE(starts[w] >= 0 && lens[w] >= 0 && starts[w] + lens[w] <= n_samples, WB_ERR_ARG,
               "window %d [%lld, +%lld) outside the waveform (%lld samples)", w, (long long)starts[w],
               (long long)lens[w], (long long)n_samples);
    WB_REQUIRE(lens[w] >= MEL_N_FFT, WB_ERR_SHAPE, "window %d has %lld samples < n_fft = 400 (audio.rs:292)", w,
               (long long)lens[w]);
    WB_REQUIRE(lens[w] < ((int64_t)1 << 30), WB_ERR_SHAPE, "window %d too long", w);
    const int nf = (int)(lens[w] / MEL_HOP);
    wins[w] = MelWindow{starts[w], (int32_t)lens[w], nf, std::min(nf, (int)clip_frames), 0};   // transcribe.rs:171-177
    const int T = wins[w].n_emit + padding;
    if (frames_out) frames_out[w] = T;
    maxF = std::max(maxF, nf); maxT = std::max(maxT, T);
  }
  WB_REQUIRE(row_stride >= maxT && row_stride % 4 == 0 && win_stride >= (int64_t)80 * row_stride, WB_ERR_ARG,
             "wb_waveform_to_mels_dev: row_stride %d must be a multiple of 4 and >= %d frames", row_stride, maxT);
  wb::GpuTurn turn(device);
  WB_HIP(hipSetDevice(device));
  const MelTables* tabs;
  WB_TRY(get_mel_tables(device, sample_rate, &tabs));
  DevMem d_win, d_max;
  WB_TRY(d_win.alloc(wins.size() * sizeof(MelWindow)));
  WB_TRY(d_max.alloc((size_t)n_windows * mel_bmax_stride(maxF) * 2 * 4));
  hipStream_t st = nullptr;
  WB_HIP(hipMemcpyAsync(d_win.p, wins.data(), wins.size() * sizeof(MelWindow), hipMemcpyHostToDevice, st));
  struct Events {   // released on every return path
    hipEvent_t a = nullptr, b = nullptr;
    ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  } ev;
  hipEvent_t& e0 = ev.a; hipEvent_t& e1 = ev.b;
  if (elapsed_ms) { WB_HIP(hipEventCreate(&e0)); WB_HIP(hipEventCreate(&e1)); WB_HIP(hipEventRecord(e0, st)); }
  for (int it = 0; it < iters; it++) {
    launch_mel_spectrogram(st, pcm_dev, d_win.as<MelWindow>(), n_windows, maxF, tabs, mel_dev, win_stride, row_stride,
                           d_max.as<float>(), padding, row_stride);
    launch_mel_finalize(st, d_win.as<MelWindow>(), n_windows, mel_dev, win_stride, row_stride, d_max.as<float>(), maxF);
  }
  if (elapsed_ms) WB_HIP(hipEventRecord(e1, st));
  WB_HIP(hipGetLastError());
  WB_HIP(hipStreamSynchronize(st));
  if (elapsed_ms) {
    float ms = 0.f;
    WB_HIP(hipEventElapsedTime(&ms, e0, e1));
    *elapsed_ms = ms;
  }
  return WB_OK;
}

int wb_forward_encoder(wb_model* m, const float* mel, int B, int T, float* out) {
  WB_REQUIRE(m && mel && out && B > 0, WB_ERR_ARG, "wb_forward_encoder: bad argument");
  // mod.rs:236-241
  WB_REQUIRE(T >= 1 && T <= m->max_mel_frames(), WB_ERR_SHAPE, "Audio length %d cannot exceed %d.", T,
             m->max_mel_frames());
  wb::GpuTurn turn(m->device);
  std::lock_guard<std::mutex> lk(g_stateless_mu);
  WB_HIP(hipSetDevice(m->device));
  wb_model* sc = m;
  hipStream_t st = m->stream;
  const int d = m->dims.n_audio_state, C = (T - 1) / 2 + 1;
  // device mel rows are padded to a multiple of 4 floats
  const int Ts = (T + 3) & ~3;
  WB_TRY(sc->io_a.ensure((size_t)B * 80 * Ts * 4));
  WB_TRY(sc->io_b.ensure((size_t)B * C * d * 4));
  WB_HIP(hipMemcpy2DAsync(sc->io_a.p, (size_t)Ts * 4, mel, (size_t)T * 4, (size_t)T * 4, (size_t)B * 80,
                          hipMemcpyHostToDevice, st));
  MelBatch mb;
  mb.mel = sc->io_a.as<float>(); mb.win_stride = (int64_t)80 * Ts; mb.row_stride = Ts; mb.T.assign(B, T);
  EncoderOut eo;
  WB_TRY(run_encoder(m, st, sc->ws, mb, sc->io_b.as<float>(), &eo));
  WB_HIP(hipMemcpyAsync(out, sc->io_b.p, (size_t)B * C * d * 4, hipMemcpyDeviceToHost, st));
  WB_HIP(hipStreamSynchronize(st));
  return WB_OK;
}

static int decoder_common(wb_model* m, wb_model* sc, hipStream_t st, const int32_t* tokens, int n, int L,
                          const float* enc_dev, int C, float* logits) {
  const int V = m->dims.n_vocab;
  // mod.rs:134-139
  WB_REQUIRE(L >= 1 && L <= m->dims.n_text_ctx, WB_ERR_SHAPE, "Token sequence length %d must not exceed %d.", L,
             m->dims.n_text_ctx);
  for (int64_t i = 0; i < (int64_t)n * L; i++)
    WB_REQUIRE(tokens[i] >= 0 && tokens[i] < V, WB_ERR_ARG, "token id %d out of range [0,%d)", tokens[i], V);
  const size_t tok_bytes = ((size_t)n * L * 4 + 255) & ~(size_t)255;
  WB_TRY(sc->io_c.ensure(tok_bytes + (size_t)n * L * V * 4));
  int32_t* tok_dev = sc->io_c.as<int32_t>();
  float* logits_dev = reinterpret_cast<float*>(sc->io_c.as<char>() + tok_bytes);
  WB_HIP(hipMemcpyAsync(tok_dev, tokens, (size_t)n * L * 4, hipMemcpyHostToDevice, st));
  WB_TRY(run_decoder_stateless(m, st, sc->ws, tok_dev, n, L, enc_dev, C, logits_dev));
  WB_HIP(hipMemcpyAsync(logits, logits_dev, (size_t)n * L * V * 4, hipMemcpyDeviceToHost, st));
  WB_HIP(hipStreamSynchronize(st));
  return WB_OK;
}

int wb_forward_decoder(wb_model* m, const int32_t* tokens, int n, int L, const float* enc, int C, float* logits) {
  WB_REQUIRE(m && tokens && enc && logits && n > 0 && C > 0, WB_ERR_ARG, "wb_forward_decoder: bad argument");
  wb::GpuTurn turn(m->device);
  std::lock_guard<std::mutex> lk(g_stateless_mu);
  WB_HIP(hipSetDevice(m->device));
  wb_model* sc = m;
  hipStream_t st = m->stream;
  const int d = m->dims.n_text_state;
  WB_TRY(sc->io_b.ensure((size_t)n * C * d * 4));
  WB_HIP(hipMemcpyAsync(sc->io_b.p, enc, (size_t)n * C * d * 4, hipMemcpyHostToDevice, st));
  return decoder_common(m, sc, st, tokens, n, L, sc->io_b.as<float>(), C, logits);
}

int wb_forward(wb_model* m, const float* mel, int B, int T, const int32_t* tokens, int L, float* logits) {
  WB_REQUIRE(m && mel && tokens && logits && B > 0, WB_ERR_ARG, "wb_forward: bad argument");
  WB_REQUIRE(T >= 1 && T <= m->max_mel_frames(), WB_ERR_SHAPE, "Audio length %d cannot exceed %d.", T,
             m->max_mel_frames());
  wb::GpuTurn turn(m->device);
  std::lock_guard<std::mutex> lk(g_stateless_mu);
  WB_HIP(hipSetDevice(m->device));
  wb_model* sc = m;
  hipStream_t st = m->stream;
  const int d = m->dims.n_audio_state, C = (T - 1) / 2 + 1, Ts = (T + 3) & ~3;
  WB_TRY(sc->io_a.ensure((size_t)B * 80 * Ts * 4));
  WB_TRY(sc->io_b.ensure((size_t)B * C * d * 4));
  WB_HIP(hipMemcpy2DAsync(sc->io_a.p, (size_t)Ts * 4, mel, (size_t)T * 4, (size_t)T * 4, (size_t)B * 80,
                          hipMemcpyHostToDevice, st));
  MelBatch mb;
  mb.mel = sc->io_a.as<float>(); mb.win_stride = (int64_t)80 * Ts; mb.row_stride = Ts; mb.T.assign(B, T);
  EncoderOut eo;
  WB_TRY(run_encoder(m, st, sc->ws, mb, sc->io_b.as<float>(), &eo));
  return decoder_common(m, sc, st, tokens, B, L, sc->io_b.as<float>(), C, logits);
}

}  // extern "C"
