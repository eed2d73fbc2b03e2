// Stateful session: one mel + encoder + cross-K/V pass per batch of windows, then KV-cached decode
// steps -- the result-equivalent fast path behind /root/reference/src/transcribe.rs:148-312.
#include "session.h"

#include <climits>
#include <cstring>
#include <mutex>

#include "handoff.h"

using namespace wb;

namespace wb {

Profile& profile() {
  static Profile p;
  return p;
}

ScopedTimer::ScopedTimer(hipStream_t s, int slot_) : st(s), slot(slot_), on(profile().on) {
  if (!on) return;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
  (void)hipEventRecord(a, st);
}
void ScopedTimer::stop() {
  if (on) (void)hipEventRecord(b, st);
}

// ---- per-kernel profiling --------------------------------------------------------------------------------
static const char* const g_kernel_names[KC_COUNT] = {
    "dec_prepare", "dec_attn_fused (LN + QKV + self-attention + out-proj)", "dec_cross_attn (LN + Wq + cross-attention)",
    "dec_gemv cross-attn out-proj", "dec_mlp_fused (LN + lin1 + GELU + lin2)", "dec_gemv logits (LN + E^T + tile stats)",
    "dec_topk_merge", "dec_gemv LN + QKV", "dec_self_attn", "dec_gemv self-attn out-proj", "dec_gemv LN + Wq",
    "dec_gemv LN + lin1", "dec_gemv GELU + lin2", "dec_cross_fused (LN + Wq + cross-attention + out-proj)",
    "batch: dec_resolve_ln (fold + LayerNorm)", "batch: split-K MFMA GEMM (decoder weight stream)",
    "batch: dec_self_attn (paged self-KV)", "batch: dec_cross_attn_stream (cached K/V stream)",
    "batch: dec_cross_attn chunked (cached K/V, beams)", "batch: dec_attn_combine", "batch: dec_gelu_fold",
    "batch: logits MFMA GEMM (E^T stream)", "batch: dec_topk_rows", "dec_persist (flag-chained decode steps)",
    "dec_beam_update (beam.rs bookkeeping on the device)", "dec_fold_ln_rows (final fold + LayerNorm, 9 - 16 rows)"};
struct PendingLaunch { hipEvent_t a, b; int cls; double bytes; };
static std::mutex g_prof_mu;
static std::vector<PendingLaunch> g_pending;
static KernelStat g_kstats[KC_COUNT];
static thread_local hipEvent_t tl_ev_a = nullptr, tl_ev_b = nullptr;

void prof_tag(int cls, double algo_bytes) {
  if (!profile().on) return;
  hipEvent_t a = nullptr, b = nullptr;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_pending.push_back(PendingLaunch{a, b, cls, algo_bytes});
  }
  tl_ev_a = a; tl_ev_b = b;
}
void prof_adjust_bytes(int cls, double delta) {
  if (!profile().on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_kstats[cls].bytes += delta;
}
bool prof_take_events(hipEvent_t* start, hipEvent_t* stop) {
  if (!tl_ev_a) return false;
  *start = tl_ev_a; *stop = tl_ev_b;
  tl_ev_a = tl_ev_b = nullptr;
  return true;
}
void prof_collect() {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (PendingLaunch& p : g_pending) {
    float ms = 0.f;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      g_kstats[p.cls].calls++; g_kstats[p.cls].ms += ms; g_kstats[p.cls].bytes += p.bytes;
    }
    (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b);
  }
  g_pending.clear();
  (void)hipGetLastError();      // (a tag whose launch never happened fails its elapsed-time query: not a sticky error for later calls)
}
void ScopedTimer::collect() {
  if (!on) return;
  float ms = 0.f;
  if (hipEventSynchronize(b) == hipSuccess && hipEventElapsedTime(&ms, a, b) == hipSuccess) profile().ms[slot] += ms;
}
ScopedTimer::~ScopedTimer() {
  if (a) (void)hipEventDestroy(a);
  if (b) (void)hipEventDestroy(b);
}

// sessions are recycled per model so that a transcription loop does not pay hipMalloc per batch.
// g_pool has an entry for exactly the models that are alive: a session released after its model was freed is
// destroyed instead of being parked under a dangling key, and a pooled session never outlives its model.
static std::mutex g_pool_mu;
static std::unordered_map<uint64_t, std::vector<wb_session*>> g_pool;   // by wb_model::uid (never reused), not by address
static uint64_t g_next_model_uid = 1;
constexpr size_t POOL_MAX_SESSIONS = 8;
constexpr size_t POOL_MAX_BYTES = (size_t)4 << 30;   // per model: big batches (large-v2 x 64 windows ~ 20 GB) are not parked

void session_pool_register(wb_model* m) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  m->uid = g_next_model_uid++;
  g_pool[m->uid];
}

// wb_model_free: pooled sessions of that model die with it
void session_pool_purge(wb_model* m) {
  std::vector<wb_session*> dead;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_pool.find(m->uid);
    if (it != g_pool.end()) { dead.swap(it->second); g_pool.erase(it); }
  }
  for (wb_session* s : dead) delete s;
}

static size_t session_device_bytes(const wb_session* s) {
  size_t n = 0;
  for (const DevMem* b : {&s->pcm, &s->mel, &s->wins, &s->gmax, &s->enc_out, &s->ckv, &s->win_meta, &s->kc, &s->vc,
                          &s->tabs, &s->state, &s->x, &s->h, &s->att, &s->Pqkv, &s->Po, &s->Pq, &s->P1, &s->P2, &s->Pa, &s->Pc, &s->carec, &s->ca,
                          &s->logits, &s->tstats, &s->row_stats, &s->mask, &s->lp_tmp, &s->gctl, &s->gtok, &s->hm,
                          &s->ws.x1, &s->ws.x, &s->ws.h, &s->ws.qkv, &s->ws.att, &s->ws.hm, &s->ws.desc1, &s->ws.desc2,
                          &s->ws.auxidx, &s->ws.segs, &s->ws.misc})
    n += b->bytes;
  return n;
}

int session_create(wb_model* m, int n_windows, int max_beams, int padding, wb_session** out) {
  WB_REQUIRE(m && out, WB_ERR_ARG, "session: null argument");
  WB_REQUIRE(n_windows >= 1, WB_ERR_ARG, "session: n_windows must be >= 1");
  WB_REQUIRE(max_beams >= 1 && max_beams <= MAX_BEAMS, WB_ERR_ARG, "session: max_beams must be in [1, %d]", MAX_BEAMS);
  WB_REQUIRE(padding >= 0 && padding < m->max_mel_frames(), WB_ERR_ARG, "session: bad padding %d", padding);
  WB_HIP(hipSetDevice(m->device));
  wb_session* s = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_pool.find(m->uid);
    WB_REQUIRE(it != g_pool.end(), WB_ERR_ARG, "session: the model handle is not alive");
    if (!it->second.empty()) { s = it->second.back(); it->second.pop_back(); }
  }
  if (!s) {
    s = new wb_session();
    s->m = m;
    s->model_uid = m->uid;
    s->device = m->device;
    hipError_t e = hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking);
    if (e != hipSuccess) { delete s; set_error("hipStreamCreate failed: %s", hipGetErrorString(e)); return WB_ERR_HIP; }
    e = hipHostMalloc((void**)&s->guard_host, 64, hipHostMallocMapped);
    if (e == hipSuccess) { memset(s->guard_host, 0, 64); e = hipHostGetDevicePointer((void**)&s->guard_dev, s->guard_host, 0); }
    if (e != hipSuccess) { delete s; set_error("session guard words: %s", hipGetErrorString(e)); return WB_ERR_HIP; }
  } else if (s->enc_guard_pending || s->guard_host[0] || s->guard_host[1]) {
    // a previous user left an unresolved guard behind (its call failed before the check): drain and clear
    (void)hipStreamSynchronize(s->st);
    s->guard_host[0] = s->guard_host[1] = 0;
  }
  s->enc_guard_pending = false;
  s->W = n_windows; s->max_beams = max_beams; s->S = n_windows * max_beams; s->padding = padding;
  s->T.assign(n_windows, 0); s->C.clear(); s->row0.clear();
  s->prev_len.clear(); s->prev_win.clear(); s->prev_n = 0; s->step = 0;
  s->has_mask = false; s->decode_ready = false; s->last_had_logits = 0; s->last_use_mask = 0;
  s->sample_rate = 16000.0;          // per-use state: a pooled session must not remember its previous caller
  *out = s;
  return WB_OK;
}

static int session_finish_encode(wb_session* s, const MelBatch& mb, bool defer_guard = false) {
  wb_model* m = s->m;
  const int d = m->dims.n_audio_state, NL = m->dims.n_text_layer;
  int rows = 0;
  for (int t : mb.T) rows += (t - 1) / 2 + 1;
  WB_TRY(s->enc_out.ensure((size_t)rows * d * 4));
  EncoderOut eo;
  const int rows_all = rows;
  const int ldkv_all = NL * 2 * d;
  WB_TRY(s->ckv.ensure((size_t)rows_all * ldkv_all * 4));
  // encoder + cross-K/V projection under ONE range guard of the split-precision kernel (engine.cpp: split_guarded), on
  // this session's own flag word; deferred: no synchronisation here, the decode's own resolves it
  bool deferred = false;
  WB_TRY(split_guarded(m, s->st, s->guard_host, s->guard_dev, defer_guard && !profile().on ? &deferred : nullptr, [&]() -> int {
  {
    ScopedTimer tm(s->st, 1);
    WB_TRY(run_encoder_unguarded(m, s->st, s->ws, mb, s->enc_out.as<float>(), &eo));
    tm.stop();
    if (tm.on) { WB_HIP(hipStreamSynchronize(s->st)); tm.collect(); }
  }
  {
    ScopedTimer tm(s->st, 2);
    GemmArgs g;
    // output layer-major: [layer][packed encoder row][K | V] -- one layer's cached K/V is one dense region (ckv_of)
    g.A = s->enc_out.as<float>(); g.lda = d; g.B = m->ckv_all.w; g.ldb = ldkv_all; g.C = s->ckv.as<float>(); g.ldc = 2 * d;
    g.c_block_cols = 2 * d; g.c_block_stride = (int64_t)rows_all * 2 * d;
    g.bias = m->ckv_all.b; g.M = rows_all; g.N = ldkv_all; g.K = d;
    g.col_scale = m->qk_scale; g.col_scale_period = 2 * d; g.col_scale_width = d;   // K * s (mod.rs:510-514)
    WB_TRY(gemm_dispatch(m, s->st, g, m->ckv_all.k, m->ckv_all.sh, m->ckv_all.sl));
    tm.stop();
    if (tm.on) { WB_HIP(hipStreamSynchronize(s->st)); tm.collect(); }
  }
  return WB_OK;
  }));
  s->enc_guard_pending = deferred;
  if (deferred) s->enc_mb = mb;
  s->C = eo.C; s->row0 = eo.row0; s->enc_rows = eo.rows;
  s->maxC = 0;
  for (int c : s->C) s->maxC = std::max(s->maxC, c);
  s->n_chunks = (s->maxC + cross_attn_chunk() - 1) / cross_attn_chunk();
  WB_REQUIRE(s->n_chunks <= CA_NCH_MAX, WB_ERR_SHAPE, "encoder context %d too long for the decode kernels", s->maxC);
  // (cross-attention K|V of every decoder layer, once per window -- mod.rs:484-485 does it per layer / beam / step -- ran
  // above, under the encoder's range guard)
  std::vector<int> meta(2 * s->W);
  for (int w = 0; w < s->W; w++) { meta[w] = s->row0[w]; meta[s->W + w] = s->C[w]; }
  WB_TRY(s->win_meta.ensure(meta.size() * 4));
  if (meta != s->meta_host || s->meta_dev_ptr != s->win_meta.p) {   // (same geometry as last time: already on the device)
    s->meta_host.clear(); s->meta_dev_ptr = nullptr;                 // void the cache key before the contents change
    WB_HIP(hipMemcpyAsync(s->win_meta.p, meta.data(), meta.size() * 4, hipMemcpyHostToDevice, s->st));
    WB_HIP(hipStreamSynchronize(s->st));
    s->meta_host = meta; s->meta_dev_ptr = s->win_meta.p;
  }
  return WB_OK;
}

void session_rewind(wb_session* s) {
  s->prev_len.clear(); s->prev_win.clear(); s->prev_n = 0; s->step = 0;
  s->last_had_logits = 0; s->last_use_mask = 0; s->prof_step_off = 0;
}

// After s->st was synchronised behind a deferred encode pass: if the split-precision kernel raised this session's flag, the
// model has left that kernel and the pass is repeated here on the exact-f32 one (*reencoded = true: the encoder output and
// the cached cross K/V changed under whatever was decoded from them -- decode again).
int session_enc_guard_resolve(wb_session* s, bool* reencoded) {
  *reencoded = false;
  if (!s->enc_guard_pending) return WB_OK;
  s->enc_guard_pending = false;
  if (!split_guard_resolve(s->m, s->guard_host)) return WB_OK;
  *reencoded = true;
  return session_finish_encode(s, s->enc_mb, false);
}

int session_encode_pcm(wb_session* s, const float* pcm, int64_t n_pcm, const int64_t* starts, const int64_t* lens,
                       bool pcm_on_device, bool defer_guard) {
  wb_model* m = s->m;
  const int clip = m->max_mel_frames() - s->padding;   // transcribe.rs:171-177
  int64_t lo = n_pcm, hi = 0;
  for (int w = 0; w < s->W; w++) {
    WB_REQUIRE(starts[w] >= 0 && lens[w] >= 0 && starts[w] + lens[w] <= n_pcm, WB_ERR_ARG,
               "window %d [%lld, +%lld) outside the waveform (%lld samples)", w, (long long)starts[w],
               (long long)lens[w], (long long)n_pcm);
    WB_REQUIRE(lens[w] >= MEL_N_FFT, WB_ERR_SHAPE, "window %d has %lld samples < n_fft = 400 (audio.rs:292)", w,
               (long long)lens[w]);
    WB_REQUIRE(lens[w] < ((int64_t)1 << 30), WB_ERR_SHAPE, "window %d too long", w);
    lo = std::min(lo, starts[w]); hi = std::max(hi, starts[w] + lens[w]);
  }
  std::vector<MelWindow> wins(s->W);
  int maxF = 0, maxT = 0;
  for (int w = 0; w < s->W; w++) {
    const int nf = (int)(lens[w] / MEL_HOP);
    wins[w] = MelWindow{pcm_on_device ? starts[w] : starts[w] - lo, (int32_t)lens[w], nf, std::min(nf, clip), 0};
    s->T[w] = wins[w].n_emit + s->padding;
    maxF = std::max(maxF, nf); maxT = std::max(maxT, s->T[w]);
  }
  const int Ts = (maxT + 3) & ~3;
  const MelTables* tabs;
  WB_TRY(get_mel_tables(m->device, s->sample_rate, &tabs));
  if (!pcm_on_device) WB_TRY(s->pcm.ensure((size_t)(hi - lo) * 4));
  WB_TRY(s->wins.ensure(wins.size() * sizeof(MelWindow)));
  WB_TRY(s->gmax.ensure((size_t)s->W * mel_bmax_stride(maxF) * 2 * 4));
  WB_TRY(s->mel.ensure((size_t)s->W * 80 * Ts * 4));
  static const bool stage_pcm = []() { const char* e = getenv("WHISPER_HIP_PCM_STAGE"); return e && e[0] == '1'; }();
  if (!pcm_on_device && stage_pcm) {
    const size_t nb = (size_t)(hi - lo) * 4;
    if (s->pcm_stage_bytes < nb) {
      if (s->pcm_stage) (void)hipHostFree(s->pcm_stage);
      s->pcm_stage = nullptr; s->pcm_stage_bytes = 0;
      WB_HIP(hipHostMalloc((void**)&s->pcm_stage, nb, hipHostMallocDefault));
      s->pcm_stage_bytes = nb;
    }
    memcpy(s->pcm_stage, pcm + lo, nb);
    WB_HIP(hipMemcpyAsync(s->pcm.p, s->pcm_stage, nb, hipMemcpyHostToDevice, s->st));
  } else
  if (!pcm_on_device) WB_HIP(hipMemcpyAsync(s->pcm.p, pcm + lo, (size_t)(hi - lo) * 4, hipMemcpyHostToDevice, s->st));
  const float* pcm_dev = pcm_on_device ? pcm : s->pcm.as<float>();
  // the window table on the device is re-used when this (pooled) session saw the same windows last time
  const bool same_wins = s->wins_host.size() == wins.size() && s->wins_dev_ptr == s->wins.p &&
                         memcmp(s->wins_host.data(), wins.data(), wins.size() * sizeof(MelWindow)) == 0;
  if (!same_wins) {
    s->wins_host.clear(); s->wins_dev_ptr = nullptr;                  // void the cache key before the contents change
    WB_HIP(hipMemcpyAsync(s->wins.p, wins.data(), wins.size() * sizeof(MelWindow), hipMemcpyHostToDevice, s->st));
    WB_HIP(hipStreamSynchronize(s->st));   // `wins` is a stack vector
    s->wins_host = wins; s->wins_dev_ptr = s->wins.p;
  } else if (!pcm_on_device) {
    WB_HIP(hipStreamSynchronize(s->st));   // the caller's PCM buffer may go away
  }
  static const int trace_extra = []() { const char* e = getenv("WHISPER_HIP_ENC_TRACE_EXTRA"); return e ? atoi(e) : 0; }();
  if (!pcm_on_device && (trace_extra & 1)) enc_trace_stage(s->st, "pcm", s->pcm.p, (size_t)(hi - lo) * 4);
  {
    ScopedTimer tm(s->st, 0);
    launch_mel_spectrogram(s->st, pcm_dev, s->wins.as<MelWindow>(), s->W, maxF, tabs, s->mel.as<float>(),
                           (int64_t)80 * Ts, Ts, s->gmax.as<float>(), s->padding, Ts);
    if (trace_extra & 2) enc_trace_stage(s->st, "mel0", s->mel.p, (size_t)s->W * 80 * Ts * 4);
    if (trace_extra & 4) enc_trace_stage(s->st, "gmax", s->gmax.p, (size_t)s->W * mel_bmax_stride(maxF) * 2 * 4);
    launch_mel_finalize(s->st, s->wins.as<MelWindow>(), s->W, s->mel.as<float>(), (int64_t)80 * Ts, Ts,
                        s->gmax.as<float>(), maxF);
    tm.stop();
    if (tm.on) { WB_HIP(hipStreamSynchronize(s->st)); tm.collect(); profile().ms[5] += 1; }
  }
  WB_HIP(hipGetLastError());
  MelBatch mb;
  mb.mel = s->mel.as<float>(); mb.win_stride = (int64_t)80 * Ts; mb.row_stride = Ts; mb.T = s->T;
  return session_finish_encode(s, mb, defer_guard);
}

int session_reserve(wb_session* s, int max_len) {
  wb_model* m = s->m;
  const wb_dims& D = m->dims;
  const int d = D.n_text_state, NL = D.n_text_layer, V = D.n_vocab, S = s->S;
  WB_REQUIRE(d <= 1280, WB_ERR_SHAPE, "decode kernels keep whole rows in LDS: n_state %d > 1280", d);
  WB_REQUIRE(d % 64 == 0, WB_ERR_SHAPE, "decode kernels need n_state %% 64 == 0 (head size 64), got %d", d);
  max_len = std::max(8, std::min(max_len, std::min(D.n_text_ctx, 448)));
  s->Lmax = max_len;
  const size_t pool = (size_t)max_len * S;
  WB_TRY(s->kc.ensure_zeroed((size_t)NL * pool * d * 4, s->st));
  WB_TRY(s->vc.ensure_zeroed((size_t)NL * pool * d * 4, s->st));
  WB_TRY(s->tabs.ensure((size_t)2 * S * max_len * 4));
  s->lay = make_step_layout(S, s->W);
  WB_TRY(s->state.ensure((size_t)s->lay.total * 4));
  // step state and top-k results live in mapped pinned host memory: the prepare kernel pulls the
  // state over PCIe, the merge kernel pushes the k (id, log-prob) pairs -- no copy launches per step
  const size_t host_bytes = (size_t)s->lay.total * 4 + (size_t)S * TOPK_MAX * 8 + (size_t)S * 2 * 4;   // + chain flags
  if (s->host_bytes < host_bytes) {
    if (s->host_block) { (void)hipHostFree(s->host_block); s->host_block = nullptr; s->host_bytes = 0; }
    WB_HIP(hipHostMalloc((void**)&s->host_block, host_bytes, hipHostMallocMapped));
    s->host_bytes = host_bytes;
  }
  WB_HIP(hipHostGetDevicePointer((void**)&s->host_block_dev, s->host_block, 0));
  s->state_host = reinterpret_cast<int*>(s->host_block);
  s->topk_id_host = reinterpret_cast<int32_t*>(s->host_block + (size_t)s->lay.total * 4);
  s->topk_lp_host = reinterpret_cast<float*>(s->host_block + (size_t)s->lay.total * 4 + (size_t)S * TOPK_MAX * 4);
  s->chain_flags_off = (size_t)s->lay.total * 4 + (size_t)S * TOPK_MAX * 8;
  gemv_plan(d, 3 * d, &s->ks_qkv, &s->ksl_qkv);
  gemv_plan(d, d, &s->ks_o, &s->ksl_o);
  gemv_plan(d, 4 * d, &s->ks_1, &s->ksl_1);
  gemv_plan(4 * d, d, &s->ks_2, &s->ksl_2);
  s->ks_v = 1; s->ksl_v = d;                    // logits: whole rows per block (tile statistics need complete sums)
  s->ct_v = GV_CT_LOGITS;
  s->n_tiles_v = (V + s->ct_v - 1) / s->ct_v;
  // (+ 8 rows of slack: the fused kernels read whole MR-row tiles unconditionally, live or not)
  WB_TRY(s->x.ensure(((size_t)2 * S + 8) * d * 4));   // residual stream, ping-pong
  WB_TRY(s->h.ensure((size_t)S * d * 4));
  WB_TRY(s->att.ensure((size_t)S * d * 4));
  WB_TRY(s->hm.ensure((size_t)S * 4 * d * 4));
  // (batch mode's skinny GEMM, decode_batch.hip, splits K its own way: the plane buffers hold the larger count)
  s->sk_qkv = skinny_ksplit(d, 3 * d, KS_MAX, S); s->sk_o = skinny_ksplit(d, d, KS_MAX, S);
  s->sk_1 = skinny_ksplit(d, 4 * d, KS_MAX, S); s->sk_2 = skinny_ksplit(4 * d, d, KS_MAX, S);
  WB_TRY(s->Pqkv.ensure((size_t)std::max(s->ks_qkv, s->sk_qkv) * S * 3 * d * 4));
  WB_TRY(s->Po.ensure(((size_t)std::max(s->ks_o, s->sk_o) * S + 8) * d * 4));
  WB_TRY(s->Pq.ensure((size_t)std::max(s->ks_o, s->sk_o) * S * d * 4));
  WB_TRY(s->P1.ensure((size_t)std::max(s->ks_1, s->sk_1) * S * 4 * d * 4));
  WB_TRY(s->P2.ensure(((size_t)std::max(std::max(s->ks_2, s->sk_2), dec_mlp_fused_planes(d)) * S + 8) * d * 4));

  WB_TRY(s->Pa.ensure(((size_t)D.n_text_head * S + 8) * d * 4));
  WB_TRY(s->Pc.ensure(((size_t)D.n_text_head * S + 8) * d * 4));
  WB_TRY(s->carec.ensure(((size_t)D.n_text_head * std::max(1, s->n_chunks) * S + 8) * (d + 2) * 4));
  WB_TRY(s->ca.ensure((size_t)S * D.n_text_head * std::max(1, s->n_chunks) * CA_STRIDE * 4));
  WB_TRY(s->logits.ensure((size_t)S * V * 4));
  WB_TRY(s->tstats.ensure((size_t)S * s->n_tiles_v * TS_STRIDE * 4));
  WB_TRY(s->row_stats.ensure((size_t)S * 2 * 4));
  WB_TRY(s->lp_tmp.ensure((size_t)V * 4));
  s->decode_ready = true;
  return WB_OK;
}

}  // namespace wb

void wb_session::clear_graphs() {
  for (auto& kv : graphs) (void)hipGraphExecDestroy(kv.second);
  graphs.clear();
}

wb_session::~wb_session() {
  clear_graphs();
  if (host_block) (void)hipHostFree(host_block);
  if (guard_host) (void)hipHostFree(guard_host);
  if (pcm_stage) (void)hipHostFree(pcm_stage);
  if (ev_seg) (void)hipEventDestroy(ev_seg);
  if (st2) (void)hipStreamDestroy(st2);
  if (st) (void)hipStreamDestroy(st);
}

extern "C" {

int wb_session_begin(wb_model* m, const float* pcm, int64_t n_pcm, const int64_t* starts, const int64_t* lens,
                     int n_windows, int max_beams, int padding, wb_session** out) {
  WB_REQUIRE(m && pcm && starts && lens && out, WB_ERR_ARG, "wb_session_begin: null argument");
  wb::GpuTurn turn(m->device);
  wb_session* s = nullptr;
  WB_TRY(session_create(m, n_windows, max_beams, padding, &s));
  int rc = session_encode_pcm(s, pcm, n_pcm, starts, lens, false);
  if (rc != WB_OK) { wb_session_free(s); return rc; }
  *out = s;
  return WB_OK;
}

int wb_session_begin_mel(wb_model* m, const float* mel, const int32_t* T, int n_windows, int max_beams, int padding,
                         wb_session** out) {
  WB_REQUIRE(m && mel && T && out, WB_ERR_ARG, "wb_session_begin_mel: null argument");
  wb::GpuTurn turn(m->device);
  wb_session* s = nullptr;
  WB_TRY(session_create(m, n_windows, max_beams, padding, &s));
  const int clip = m->max_mel_frames() - padding;
  int maxT = 0;
  for (int w = 0; w < n_windows; w++) {
    if (T[w] < 1) { wb_session_free(s); set_error("window %d: empty mel", w); return WB_ERR_SHAPE; }
    s->T[w] = std::min(T[w], clip) + padding;   // transcribe.rs:171-177
    maxT = std::max(maxT, s->T[w]);
  }
  const int Ts = (maxT + 3) & ~3;
  std::vector<float> host((size_t)n_windows * 80 * Ts, 0.f);
  const float* src = mel;
  for (int w = 0; w < n_windows; w++) {
    const int keep = s->T[w] - padding;
    for (int r = 0; r < 80; r++) memcpy(&host[((size_t)w * 80 + r) * Ts], src + (size_t)r * T[w], (size_t)keep * 4);
    src += (size_t)80 * T[w];
  }
  int rc = s->mel.ensure(host.size() * 4);
  if (rc == WB_OK && (hipMemcpyAsync(s->mel.p, host.data(), host.size() * 4, hipMemcpyHostToDevice, s->st) != hipSuccess ||
                      hipStreamSynchronize(s->st) != hipSuccess)) {
    set_error("mel upload failed");
    rc = WB_ERR_HIP;
  }
  if (rc == WB_OK) {
    MelBatch mb;
    mb.mel = s->mel.as<float>(); mb.win_stride = (int64_t)80 * Ts; mb.row_stride = Ts; mb.T = s->T;
    rc = session_finish_encode(s, mb);
  }
  if (rc != WB_OK) { wb_session_free(s); return rc; }
  *out = s;
  return WB_OK;
}

void wb_session_free(wb_session* s) {
  if (!s) return;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_pool.find(s->model_uid);   // absent: the model was freed first -- s->m dangles and must not be touched
                                            // (a NEW model at the same address has another uid: no false match)
    if (it != g_pool.end()) {
      size_t parked = 0;
      for (const wb_session* q : it->second) parked += session_device_bytes(q);
      // keep the allocations (and captured graphs) for the next batch, within a byte budget
      static const bool pool_enabled = []() { const char* e = getenv("WHISPER_HIP_SESSION_POOL"); return !(e && e[0] == '0'); }();   // developer A/B
      if (pool_enabled && it->second.size() < POOL_MAX_SESSIONS && parked + session_device_bytes(s) <= POOL_MAX_BYTES) {
        it->second.push_back(s);
        return;
      }
    }
  }
  (void)hipSetDevice(s->device);
  delete s;
}

int wb_session_set_special_mask(wb_session* s, const uint8_t* is_special) {
  WB_REQUIRE(s && is_special, WB_ERR_ARG, "wb_session_set_special_mask: null argument");
  wb::GpuTurn turn(s->device);
  const int V = s->m->dims.n_vocab;
  WB_HIP(hipSetDevice(s->m->device));
  WB_TRY(s->mask.ensure((size_t)V * 4));
  // a pooled session that already holds this mask on the device (the usual case: one tokenizer per process) skips the
  // 200 KB upload and its synchronisation -- ~40 us of every wb_waveform_to_tokens call
  if (s->mask_dev_ptr == s->mask.p && (int)s->mask_host.size() == V && memcmp(s->mask_host.data(), is_special, (size_t)V) == 0) {
    s->has_mask = true;
    return WB_OK;
  }
  s->mask_host.clear(); s->mask_dev_ptr = nullptr;                       // void the cache key before the contents change
  std::vector<float> mk(V);
  for (int i = 0; i < V; i++) mk[i] = is_special[i] ? -INFINITY : 0.f;   // transcribe.rs:244
  // (on the session's own stream: a copy on the legacy stream fails while ANOTHER session of the process is capturing a
  // step graph -- "would make the legacy stream depend on a capturing blocking stream", profiles/r06_b_two_lanes.txt)
  WB_HIP(hipMemcpyAsync(s->mask.p, mk.data(), (size_t)V * 4, hipMemcpyHostToDevice, s->st));
  WB_HIP(hipStreamSynchronize(s->st));
  s->mask_host.assign(is_special, is_special + V); s->mask_dev_ptr = s->mask.p;
  s->has_mask = true;
  return WB_OK;
}

// Enqueue the kernels of one decode step on `st`.  Everything that changes from step to step (token
// ids, parents, lengths, the live-beam count, table parity) is read by the kernels from the step
// state, so for a given (row bucket, k, mask, fuse) the launch sequence is identical every step and
// can be captured once into a hipGraph and replayed.
// Range guard of the split-precision decoder GEMM (decode_batch.hip): called wherever a decode has just synchronised with
// the host.  The kernel raises the model's mapped flag word when a result is not finite (an activation outside fp16's
// range, |x| >= 65504: attention outputs and GELU hidden units are the only GEMM inputs that are not LayerNorm outputs).
// The call that observes it fails loudly, the model switches to the exact-f32 skinny kernel for good and this session's
// captured step graphs are dropped, so the caller's retry decodes with f32 GEMMs.  The flag word is this SESSION's
// (guard_host[1]): only the session whose rows were invalid fails; the others keep their (finite) results and drop their
// own graphs at their next look-up, where the graph signature carries the model's switch (launch_step).
static int dec_split_check(wb_session* s) {
  wb_model* m = s->m;
  if (!s->guard_host || __atomic_load_n(&s->guard_host[1], __ATOMIC_ACQUIRE) == 0) return WB_OK;
  __atomic_store_n(&s->guard_host[1], 0, __ATOMIC_RELEASE);
  __atomic_store_n(&m->dec_split_off, 1, __ATOMIC_RELEASE);
  s->clear_graphs();
  WB_REQUIRE(false, WB_ERR_STATE, "a decoder activation left fp16's range under the split-precision decode GEMM: this call's "
             "rows are invalid; the model now uses the exact-f32 decoder GEMMs -- decode again");
  return WB_OK;
}

// device-chained beam search: where a step reads its state block and leaves its top-k rows, and the bookkeeping launch behind it
struct BeamStepIO { const int* state_src; int32_t* topk_id; float* topk_lp; BeamChainArgs upd; };

static int enqueue_step(wb_session* s, int n_launch, int k, int use_mask, bool fuse_ln, int max_nb, bool timed,
                        bool chained = false, int eot = -1, const BeamStepIO* bio = nullptr) {
  wb_model* m = s->m;
  const wb_dims& D = m->dims;
  const int d = D.n_text_state, H = D.n_text_head, NL = D.n_text_layer, V = D.n_vocab, S = s->S;
  const StepLayout& L = s->lay;
  hipStream_t st = s->st;
  const int* hst = bio ? bio->state_src : reinterpret_cast<const int*>(s->host_block_dev);   // mapped view of state_host
  int32_t* out_id_dev = bio ? bio->topk_id : reinterpret_cast<int32_t*>(s->host_block_dev + (size_t)L.total * 4);
  float* out_lp_dev = bio ? bio->topk_lp : reinterpret_cast<float*>(s->host_block_dev + (size_t)L.total * 4 + (size_t)S * TOPK_MAX * 4);
  const int* dst = s->state.as<int>();
  int* tabs = s->tabs.as<int>();
  float* xb[2] = {s->x.as<float>(), s->x.as<float>() + (size_t)S * d};
  int xi = 0;                                     // xb[xi] holds the current residual stream
  float *h = s->h.as<float>(), *att = s->att.as<float>();
  const size_t pool = (size_t)s->Lmax * S;
  // cached cross K|V: layer-major, [layer][packed encoder row][2d] (session_finish_encode) -- layer l's rows start at ckv_of(l)
  const int ldkv = 2 * d;
  auto ckv_of = [&](int l) { return s->ckv.as<float>() + (size_t)l * s->enc_rows * 2 * d; };
  const int* win_row0 = s->win_meta.as<int>();
  const int* win_C = win_row0 + s->W;
  const int n = n_launch;
  // algorithmic bytes of the tagged launches (profiling): the weights a launch streams + the cached K/V it reads
  const double wsz = 4.0, dd = (double)d * d;
  double ckv_bytes = 0;
  for (int c : s->C) ckv_bytes += 8.0 * c * d;                    // K and V rows of one layer, f32
  const double self_kv_bytes = 8.0 * (double)n * (s->step + s->prof_step_off + 1) * d;

  int* gctl = chained ? s->gctl.as<int>() : nullptr;
  // chained small-batch steps: the previous step's merge kernel already prepared this one (the chain's
  // first step is prepared by session_greedy_chain)
  const bool merge_prepares = chained && fuse_ln;
  // (device-chained beam search: the bookkeeping launch behind the previous step prepared this one)
  if (!merge_prepares && !bio) prof_tag(KC_PREPARE, 8.0 * n * d);
  if (!merge_prepares && !bio)
    launch_dec_prepare(st, hst, s->state.as<int>(), L, n, tabs, s->Lmax, m->tok_emb, m->dec_pos, d, xb[0], gctl);
  auto gemv = [&](const LinearW& w, int ks, int ksl, int pro, const float* src, int ld_src, float* P) {
    GemvArgs a;
    a.W = w.w; a.ldw = w.n; a.K = w.k; a.N = w.n; a.KS = ks; a.KSL = ksl; a.pro = pro; a.src = src; a.ld_src = ld_src;
    a.P = P; a.st = dst; a.S = S;
    return a;
  };
  // y = W . LN(x + pending): folds the pending sublayer output into the residual stream (ping-pong),
  // normalises, multiplies -- in one launch when few beams are live
  auto ln_gemv = [&](GemvArgs a, const float* pend, int ks_pend, const float* pbias, const LayerNormW& ln, bool stats) {
    a.ln_g = ln.g; a.ln_b = ln.b; a.ln_eps = ln.eps; a.ln_inside = m->ln_eps_inside_sqrt;
    a.pro = PRO_LN; a.src = xb[xi]; a.ld_src = d; a.pend = pend; a.KSp = ks_pend; a.pbias = pbias; a.x_out = xb[xi ^ 1];
    launch_dec_gemv(st, a, n, stats);
    xi ^= 1;
  };
  if (!fuse_ln) {
    // ---- batch mode (more than 8 live beams): rows are many enough for the matrix cores ----
    // LayerNorm in its own launch, split-K exact-f32 MFMA GEMMs streaming each weight once into the
    // same partial-sum planes the small-batch consumers fold
    float* hm = s->hm.as<float>();
    auto big = [&](const LinearW& w, int ks, const float* A, float* P) -> int {
      GemmArgs g;
      g.A = A; g.lda = w.k; g.B = w.w; g.ldb = w.n; g.C = P; g.ldc = w.n; g.M = n; g.N = w.n; g.K = w.k;
      g.ksplit = ks; g.c_split_stride = (int64_t)S * w.n;
      prof_tag(KC_B_GEMM, wsz * (double)w.k * w.n + 4.0 * n * ((double)w.k + (double)ks * w.n));
      return gemm_dispatch(m, st, g, w.k);
    };
    auto resolve = [&](const float* pend, int ks_pend, const float* pbias, const LayerNormW& ln) {
      prof_tag(KC_B_RESOLVE_LN, 4.0 * n * d * (ks_pend + 3));
      launch_dec_resolve_ln(st, dst, n, xb[xi], xb[xi ^ 1], pend, ks_pend, S, pbias, d, ln, m->ln_eps_inside_sqrt, h);
      xi ^= 1;
    };
    // the skinny weight-stream GEMM (decode_batch.hip) for up to 64 rows of exact-f32 models: every weight row in flight
    // from the start.  WHISPER_HIP_BATCH_SKINNY=0 keeps the tiled GEMM.
    static const bool skinny_enabled = []() { const char* e = getenv("WHISPER_HIP_BATCH_SKINNY"); return !(e && e[0] == '0'); }();
    const bool skinny = skinny_enabled && n <= 64 && s->sk_qkv > 0 && s->sk_o > 0 &&
                        s->sk_1 > 0 && s->sk_2 > 0;
    const bool dec_split = m->dec_split_active();
    auto thin = [&](const LinearW& w, int ks, const float* A, float* P) -> int {
      SkinnyArgs g;
      g.A = A; g.lda = w.k; g.B = w.w; g.ldb = w.n; g.M = n; g.N = w.n; g.K = w.k; g.ksplit = ks;
      g.P = P; g.plane = S * w.n;
      if (dec_split && w.th && w.tl) { g.Bh = w.th; g.Bl = w.tl; g.range_flag = s->guard_dev + 1; g.st = dst; }   // 16-bit matrix path, f32-grade
      prof_tag(KC_B_GEMM, wsz * (double)w.k * w.n + 4.0 * n * ((double)w.k + (double)ks * w.n));
      WB_REQUIRE(launch_dec_skinny_gemm(st, g) == 0, WB_ERR_SHAPE, "skinny gemm: unsupported shape M=%d N=%d K=%d ks=%d", n,
                 w.n, w.k, ks);
      return WB_OK;
    };
    const float* pend = nullptr; int ks_pend = 0; const float* pbias = nullptr;
    // one beam per window (greedy over many windows): one block per (head, window) streams the whole cached K/V and
    // writes the normalised head outputs -- no 128-key chunk partials, no combine launch (WHISPER_HIP_CROSS_STREAM=0:
    // the chunked kernel + combine)
    static const bool cross_stream_enabled = []() { const char* e = getenv("WHISPER_HIP_CROSS_STREAM"); return !(e && e[0] == '0'); }();
    const bool cross_stream = cross_stream_enabled && max_nb <= 1;
    // ... and where the head's slice of Wq is small next to the window's cached K/V (d <= 768: `small` and below) those
    // blocks fold the pending planes, normalise and project their own query first -- two launches less per layer.
    // Measured both ways (profiles/r03_j_*): small, 10 min +3 %; large-v2 (327 KB of Wq per block, 220 VGPRs)
    // 441x -> 433x, so d = 1024 / 1280 keep the launches.  WHISPER_HIP_CROSS_STREAM_FUSE=0 / 1 forces it off / on.
    static const int stream_fuse_mode = []() { const char* e = getenv("WHISPER_HIP_CROSS_STREAM_FUSE"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
    const bool stream_fused = cross_stream && cross_stream_can_fuse(d) &&
                              (stream_fuse_mode < 0 ? d <= 768 : stream_fuse_mode == 1);
    const int kq = skinny ? s->sk_qkv : s->ks_qkv, ko = skinny ? s->sk_o : s->ks_o;
    for (int l = 0; l < NL; l++) {
      const DecBlockW& b = m->dec[l];
      resolve(pend, ks_pend, pbias, b.ln1);
      if (skinny) WB_TRY(thin(b.qkv, kq, h, s->Pqkv.as<float>()));
      else WB_TRY(big(b.qkv, kq, h, s->Pqkv.as<float>()));
      prof_tag(KC_B_SELF_ATTN, self_kv_bytes + 4.0 * n * 3 * d * kq);
      s->prof_cls_self = KC_B_SELF_ATTN;
      launch_dec_self_attn(st, dst, L, n, H, s->Pqkv.as<float>(), kq, b.qkv.b, d,
                           s->kc.as<float>() + (size_t)l * pool * d, s->vc.as<float>() + (size_t)l * pool * d, tabs,
                           s->Lmax, m->qk_scale, att);
      if (skinny) WB_TRY(thin(b.out, ko, att, s->Po.as<float>()));
      else WB_TRY(big(b.out, ko, att, s->Po.as<float>()));
      if (stream_fused) {
        // fold + cross_attn_ln + Wq inside the streaming blocks: two launches less per layer
        CaStreamFuse fz;
        fz.x_in = xb[xi]; fz.x_out = xb[xi ^ 1]; fz.pend = s->Po.as<float>(); fz.KSp = ko; fz.pbias = b.out.b;
        fz.ln_g = b.ln2.g; fz.ln_b = b.ln2.b; fz.ln_eps = b.ln2.eps; fz.ln_inside = m->ln_eps_inside_sqrt; fz.Wq = b.cq.w;
        prof_tag(KC_B_CROSS_STREAM, ckv_bytes + 4.0 * dd + 4.0 * n * d * (ko + 2));
        s->prof_cls_cross = KC_B_CROSS_STREAM;
        launch_dec_cross_attn_stream_fused(st, dst, L, s->W, H, b.cq.b, d, ckv_of(l), ldkv, 0, win_row0,
                                           win_C, m->qk_scale, att, fz);
        xi ^= 1;
      } else {
      resolve(s->Po.as<float>(), ko, b.out.b, b.ln2);
      if (skinny) WB_TRY(thin(b.cq, ko, h, s->Pq.as<float>()));
      else WB_TRY(big(b.cq, ko, h, s->Pq.as<float>()));
      if (cross_stream) {
        prof_tag(KC_B_CROSS_STREAM, ckv_bytes + 4.0 * n * d * (ko + 1));
        s->prof_cls_cross = KC_B_CROSS_STREAM;
        launch_dec_cross_attn_stream(st, dst, L, s->W, H, s->Pq.as<float>(), ko, b.cq.b, d, ckv_of(l), ldkv,
                                     0, win_row0, win_C, m->qk_scale, att);
      } else {
        prof_tag(KC_B_CROSS_CHUNK, ckv_bytes + 4.0 * n * d * ko);
        s->prof_cls_cross = KC_B_CROSS_CHUNK;
        launch_dec_cross_attn(st, dst, L, s->W, H, s->n_chunks, s->Pq.as<float>(), ko, b.cq.b, d, ckv_of(l),
                              ldkv, 0, win_row0, win_C, m->qk_scale, s->ca.as<float>(), max_nb);
        prof_tag(KC_B_COMBINE, 4.0 * n * H * s->n_chunks * CA_STRIDE);
        launch_dec_attn_combine(st, dst, n, s->ca.as<float>(), H, s->n_chunks, att);
      }
      }
      const int k1 = skinny ? s->sk_1 : s->ks_1, k2 = skinny ? s->sk_2 : s->ks_2;
      if (skinny) WB_TRY(thin(b.cout, ko, att, s->Po.as<float>()));
      else WB_TRY(big(b.cout, ko, att, s->Po.as<float>()));
      resolve(s->Po.as<float>(), ko, b.cout.b, b.ln3);
      if (skinny) WB_TRY(thin(b.mlp1, k1, h, s->P1.as<float>()));
      else WB_TRY(big(b.mlp1, k1, h, s->P1.as<float>()));
      prof_tag(KC_B_GELU_FOLD, 4.0 * n * 4 * d * (k1 + 1));
      launch_dec_gelu_fold(st, dst, n, s->P1.as<float>(), k1, S, 4 * d, b.mlp1.b, hm);
      if (skinny) WB_TRY(thin(b.mlp2, k2, hm, s->P2.as<float>()));
      else WB_TRY(big(b.mlp2, k2, hm, s->P2.as<float>()));
      pend = s->P2.as<float>(); ks_pend = k2; pbias = b.mlp2.b;
    }
    if (k > 0) {
      resolve(pend, ks_pend, pbias, m->ln_dec);
      ScopedTimer tm_logits(st, 6);
      GemmArgs g;
      g.A = h; g.lda = d; g.B = m->tok_emb_t; g.ldb = m->vocab_ld; g.C = s->logits.as<float>(); g.ldc = V;
      g.M = n; g.N = V; g.K = d;
      prof_tag(KC_B_LOGITS_GEMM, wsz * (double)V * d + 4.0 * n * ((double)d + V));
      WB_TRY(gemm_dispatch(m, st, g, d));
      tm_logits.stop();
      prof_tag(KC_B_TOPK_ROWS, 4.0 * (double)n * V);
      launch_dec_topk_rows(st, s->state.as<int>(), n, s->logits.as<float>(), V, s->mask.as<float>(), use_mask, k, out_id_dev,
                           out_lp_dev, s->row_stats.as<float>(), L, gctl, s->gtok.as<int>(), s->Lmax, eot);
      if (bio) { prof_tag(KC_BEAM_UPDATE, 8.0 * n * k); launch_dec_beam_update(st, bio->upd); }
      if (timed && tm_logits.on) {
        WB_HIP(hipStreamSynchronize(st));
        tm_logits.collect();
        prof_collect();
        profile().ms[7] += 1;
      }
    }
    return WB_OK;
  }
  // small models, exact-f32 weights, few enough blocks for one resident wave of them: the cross-attention blocks
  // project their own queries (decode.hip)
  static const bool fuse_q_enabled = []() { const char* e = getenv("WHISPER_HIP_FUSE_Q"); return !(e && e[0] == '0'); }();
  const bool fuse_q = fuse_q_enabled && fuse_ln && cross_attn_can_fuse_q(d) &&
                      s->n_chunks * H * s->W <= 256;
  // sublayer fusion (decode_fused.hip): self-attention block and MLP block are ONE launch each
  static const bool fuse_sub_enabled = []() { const char* e = getenv("WHISPER_HIP_FUSE_SUB"); return !(e && e[0] == '0'); }();
  const bool fuse_sub = fuse_sub_enabled && dec_fused_supported(d) && d == 64 * H;
  // the whole cross-attention sublayer (LN + Wq + attention over the window's cached K/V + Wo) as one launch per
  // (head, beam): 10.6 us against 9.5 + 5.9 us (+ a kernel boundary) for chunked cross-attention + out-projection GEMV
  // once the block keeps its head's whole K in flight (decode_fused.hip); WHISPER_HIP_FUSE_X=0 restores the chunked pair
  static const bool fuse_x_enabled = []() { const char* e = getenv("WHISPER_HIP_FUSE_X"); return !(e && e[0] == '0'); }();
  const bool fuse_x = fuse_x_enabled && fuse_sub && s->maxC <= CROSS_FUSED_MAX_PASSES * CROSS_FUSED_MAX_C;
  // ... or (older, opt-in) the chunked cross-attention blocks apply their head's rows of the out-projection
  // (opt-in: with 128-key chunks the MLP prologue has 6 H records per row to combine and loses what the launch saves)
  static const bool fuse_co_enabled = []() { const char* e = getenv("WHISPER_HIP_FUSE_CO"); return e && e[0] == '1'; }();
  const bool fuse_co = fuse_co_enabled && fuse_sub && fuse_q && H * s->n_chunks <= 48 && !fuse_x;
  const int nb_mlp = dec_mlp_fused_planes(d);
  const int ks_mlp = fuse_sub ? nb_mlp : s->ks_2;            // planes the MLP leaves pending
  for (int l = 0; l < NL; l++) {   // ResidualDecoderAttentionBlock::forward, mod.rs:345-350
    const DecBlockW& b = m->dec[l];
    const float* pend_a = l == 0 ? nullptr : s->P2.as<float>();
    const int ks_a = l == 0 ? 0 : ks_mlp;
    const float* pb_a = l == 0 ? nullptr : m->dec[l - 1].mlp2.b;
    const float* att_planes; int att_ks;                     // what the cross-attention prologue folds
    if (fuse_sub) {
      AttnFusedArgs fa;
      fa.st = dst; fa.lay = L; fa.S = S; fa.d = d; fa.n_head = H;
      fa.x_in = xb[xi]; fa.pend = pend_a; fa.KSp = ks_a; fa.pbias = pb_a; fa.x_out = xb[xi ^ 1];
      fa.ln_g = b.ln1.g; fa.ln_b = b.ln1.b; fa.ln_eps = b.ln1.eps; fa.ln_inside = m->ln_eps_inside_sqrt;
      fa.Wqkv = b.qkv.w; fa.ldqkv = b.qkv.n; fa.bqkv = b.qkv.b; fa.scale = m->qk_scale;
      fa.Kc = s->kc.as<float>() + (size_t)l * pool * d; fa.Vc = s->vc.as<float>() + (size_t)l * pool * d;
      fa.tabs = tabs; fa.Lmax = s->Lmax; fa.Wo = b.out.w; fa.P = s->Pa.as<float>();
      prof_tag(KC_ATTN_FUSED, 4.0 * dd * 4 + self_kv_bytes);
      s->prof_cls_self = KC_ATTN_FUSED;
      launch_dec_attn_fused(st, fa, n);
      xi ^= 1;
      att_planes = s->Pa.as<float>(); att_ks = H;
    } else {
      prof_tag(KC_GEMV_LN_QKV, wsz * dd * 3);
      ln_gemv(gemv(b.qkv, s->ks_qkv, s->ksl_qkv, PRO_PLAIN, nullptr, d, s->Pqkv.as<float>()), pend_a, ks_a, pb_a, b.ln1, false);
      prof_tag(KC_SELF_ATTN, self_kv_bytes);
      s->prof_cls_self = KC_SELF_ATTN;
      launch_dec_self_attn(st, dst, L, n, H, s->Pqkv.as<float>(), s->ks_qkv, b.qkv.b, d,
                           s->kc.as<float>() + (size_t)l * pool * d, s->vc.as<float>() + (size_t)l * pool * d, tabs,
                           s->Lmax, m->qk_scale, att);
      prof_tag(KC_GEMV_OUT, wsz * dd);
      launch_dec_gemv(st, gemv(b.out, s->ks_o, s->ksl_o, PRO_PLAIN, att, d, s->Po.as<float>()), n, false);
      att_planes = s->Po.as<float>(); att_ks = s->ks_o;
    }
    if (fuse_x) {
      CrossFusedArgs ca;
      ca.st = dst; ca.lay = L; ca.S = S; ca.d = d; ca.n_head = H;
      ca.x_in = xb[xi]; ca.pend = att_planes; ca.KSp = att_ks; ca.pbias = b.out.b; ca.x_out = xb[xi ^ 1];
      ca.ln_g = b.ln2.g; ca.ln_b = b.ln2.b; ca.ln_eps = b.ln2.eps; ca.ln_inside = m->ln_eps_inside_sqrt;
      ca.Wq = b.cq.w; ca.bq = b.cq.b; ca.scale = m->qk_scale;
      ca.ckv = ckv_of(l); ca.ldkv = ldkv; ca.koff = 0; ca.win_row0 = win_row0; ca.win_C = win_C;
      ca.Wo = b.cout.w; ca.P = s->Pc.as<float>();
      ca.n_pass = s->maxC > CROSS_FUSED_MAX_C ? 2 : 1;
      prof_tag(KC_CROSS_FUSED, ckv_bytes + 4.0 * dd * 2);
      s->prof_cls_cross = KC_CROSS_FUSED;
      launch_dec_cross_fused(st, ca, n);
      xi ^= 1;
    } else if (fuse_q) {
      // cross_attn_ln + the query projection inside the cross-attention blocks (one launch less per layer)
      CaFuse fz;
      fz.x_in = xb[xi]; fz.pend = att_planes; fz.KSp = att_ks; fz.pbias = b.out.b; fz.x_out = xb[xi ^ 1];
      fz.ln_g = b.ln2.g; fz.ln_b = b.ln2.b; fz.ln_eps = b.ln2.eps; fz.ln_inside = m->ln_eps_inside_sqrt;
      fz.Wq = b.cq.w;
      if (fuse_co) { fz.Wo = b.cout.w; fz.rec = s->carec.as<float>(); }
      prof_tag(KC_CROSS_ATTN, ckv_bytes + 4.0 * dd * (fuse_co ? 2 : 1));
      s->prof_cls_cross = KC_CROSS_ATTN;
      launch_dec_cross_attn(st, dst, L, s->W, H, s->n_chunks, nullptr, 0, b.cq.b, d, ckv_of(l), ldkv,
                            0, win_row0, win_C, m->qk_scale, s->ca.as<float>(), max_nb, &fz);
      xi ^= 1;
    } else {
      prof_tag(KC_GEMV_LN_CQ, wsz * dd);
      ln_gemv(gemv(b.cq, s->ks_o, s->ksl_o, PRO_PLAIN, nullptr, d, s->Pq.as<float>()), att_planes, att_ks, b.out.b, b.ln2,
              false);
      prof_tag(KC_CROSS_ATTN, ckv_bytes);
      s->prof_cls_cross = KC_CROSS_ATTN;
      launch_dec_cross_attn(st, dst, L, s->W, H, s->n_chunks, s->Pq.as<float>(), s->ks_o, b.cq.b, d,
                            ckv_of(l), ldkv, 0, win_row0, win_C, m->qk_scale, s->ca.as<float>(), max_nb);
    }
    if (!fuse_co && !fuse_x) {
      GemvArgs a = gemv(b.cout, s->ks_o, s->ksl_o, PRO_ATTN, s->ca.as<float>(), 0, s->Po.as<float>());
      a.n_head = H; a.n_chunks = s->n_chunks;
      prof_tag(KC_GEMV_COUT, wsz * dd);
      launch_dec_gemv(st, a, n, false);
    }
    if (fuse_sub) {
      MlpFusedArgs ma;
      ma.st = dst; ma.S = S; ma.d = d;
      ma.x_in = xb[xi]; ma.pend = s->Po.as<float>(); ma.KSp = s->ks_o; ma.pbias = b.cout.b; ma.x_out = xb[xi ^ 1];
      if (fuse_x) { ma.pend = s->Pc.as<float>(); ma.KSp = H; }
      else if (fuse_co) {   // the cross-attention blocks applied Wo themselves: fold their chunk records
        ma.pend = s->carec.as<float>(); ma.KSp = H * s->n_chunks; ma.n_head = H; ma.n_chunks = s->n_chunks;
      }
      ma.ln_g = b.ln3.g; ma.ln_b = b.ln3.b; ma.ln_eps = b.ln3.eps; ma.ln_inside = m->ln_eps_inside_sqrt;
      ma.W1 = b.mlp1.w; ma.ld1 = b.mlp1.n; ma.b1 = b.mlp1.b; ma.W2 = b.mlp2.w; ma.P = s->P2.as<float>();
      prof_tag(KC_MLP_FUSED, 4.0 * dd * 8);
      launch_dec_mlp_fused(st, ma, n);
      xi ^= 1;
    } else {
      prof_tag(KC_GEMV_LN_MLP1, wsz * dd * 4);
      ln_gemv(gemv(b.mlp1, s->ks_1, s->ksl_1, PRO_PLAIN, nullptr, d, s->P1.as<float>()), s->Po.as<float>(), s->ks_o,
              b.cout.b, b.ln3, false);
      GemvArgs a = gemv(b.mlp2, s->ks_2, s->ksl_2, PRO_GELU, s->P1.as<float>(), 4 * d, s->P2.as<float>());
      a.pbias = b.mlp1.b; a.KSp = s->ks_1;
      prof_tag(KC_GEMV_MLP2, wsz * dd * 4);
      launch_dec_gemv(st, a, n, false);
    }
  }
  if (k > 0) {
    // logits = ln(x) . token_embedding^T (mod.rs:155-156), last position only; + mask, tile statistics
    ScopedTimer tm_logits(st, 6);
    GemvArgs a;
    a.W = m->tok_emb_t; a.ldw = m->vocab_ld; a.K = d; a.N = V; a.KS = 1; a.KSL = d;
    a.P = s->logits.as<float>(); a.st = dst; a.S = S;
    a.mask = s->mask.as<float>(); a.use_mask = use_mask; a.topk = k; a.tstats = s->tstats.as<float>(); a.ct = s->ct_v;
    a.h_tmp = h;                            // (9 - 16 rows: the fold + LayerNorm runs once, in its own launch, into this buffer)
    if (dec_logits_two_launches(a, n)) prof_tag(KC_FOLD_LN_ROWS, 4.0 * n * d * (ks_mlp + 3));   // (the product launch tags itself)
    else prof_tag(KC_LOGITS, wsz * (double)V * d + 4.0 * ((double)n * d + (double)n * V));
    ln_gemv(a, s->P2.as<float>(), ks_mlp, m->dec[NL - 1].mlp2.b, m->ln_dec, true);
    tm_logits.stop();
    NextPrep nx;
    if (merge_prepares) { nx.x = xb[0]; nx.E = m->tok_emb; nx.pos = m->dec_pos; nx.tabs = tabs; nx.d = d; }
    if (chained) nx.hflags = reinterpret_cast<int*>(s->host_block_dev + s->chain_flags_off);
    prof_tag(KC_TOPK_MERGE, 4.0 * n * s->n_tiles_v * TS_STRIDE);
    launch_dec_topk_merge(st, s->state.as<int>(), n, s->tstats.as<float>(), s->n_tiles_v, k, out_id_dev, out_lp_dev,
                          s->row_stats.as<float>(), L, gctl, s->gtok.as<int>(), s->Lmax, eot, nx);
    if (bio) { prof_tag(KC_BEAM_UPDATE, 8.0 * n * k); launch_dec_beam_update(st, bio->upd); }
    if (timed && tm_logits.on) {
      WB_HIP(hipStreamSynchronize(st));
      tm_logits.collect();
      prof_collect();
      profile().ms[7] += 1;
    }
  }
  return WB_OK;
}

// Launch one decode step: replay the captured graph for this launch shape (capturing it on first use),
// or enqueue the kernels eagerly.
static int launch_step(wb_session* s, int n_launch, int k, int use_mask, bool fuse_ln, int max_nb, bool use_graph,
                       bool chained, int eot, int reps = 1, const BeamStepIO* bio = nullptr) {
  wb_model* m = s->m;
  hipStream_t st = s->st;
  if (!use_graph) {
    for (int i = 0; i < reps; i++) {
      WB_TRY(enqueue_step(s, n_launch, k, use_mask, fuse_ln, max_nb, true, chained, eot, bio));
      if (chained || bio) s->prof_step_off++;
    }
    return WB_OK;
  }
  // graphs bake in buffer addresses and launch geometry: drop them if anything moved since capture
  uint64_t sig = 1469598103934665603ull;
  auto mix = [&](uint64_t v) { sig = (sig ^ v) * 1099511628211ull; };
  for (const wb::DevMem* b : {&s->kc, &s->vc, &s->tabs, &s->state, &s->x, &s->h, &s->att, &s->Pqkv, &s->Po, &s->Pq,
                              &s->P1, &s->P2, &s->Pa, &s->Pc, &s->carec, &s->ca, &s->logits, &s->tstats, &s->row_stats, &s->mask, &s->ckv,
                              &s->win_meta, &s->gctl, &s->gtok, &s->hm, &s->bc_ctl, &s->bc_topk})
    mix((uint64_t)(uintptr_t)b->p);
  mix((uint64_t)(uintptr_t)s->host_block_dev);
  // (enc_rows: the layer-major cross-K/V cache puts layer l at ckv + l * enc_rows * 2d -- ckv_of -- so the per-layer pointers a
  // graph holds move with the batch's packed encoder rows even when no buffer does; maxC: picks the cross-attention kernel
  // and its pass count (enqueue_step);
  // dec_split_active: another session's trip switched the model's decoder GEMM under these graphs)
  for (int v : {s->S, s->W, s->Lmax, s->n_chunks, s->max_beams, m->ln_eps_inside_sqrt, eot, (int)m->dec_split_active(), s->enc_rows, s->maxC})
    mix((uint64_t)(int64_t)v);
  mix(m->uid);
  if (sig != s->buf_sig) { s->clear_graphs(); s->buf_sig = sig; }
  // (reps > 1: device-chained steps read their position from the control block, so one graph can hold
  // several consecutive steps and the host launches once per run)
  // (a beam-chain graph also bakes in the search's constants -- beam size = k, eot via the signature above, max_depth and the
  // first step's position via the control block layout: session_beam_chain drops the graphs when those change)
  const uint64_t key = ((uint64_t)reps << 48) | ((uint64_t)n_launch << 32) | ((uint64_t)k << 8) | (bio ? 8u : 0u) | (chained ? 4u : 0u) |
                       ((uint64_t)use_mask << 1) | (fuse_ln ? 1u : 0u);
  auto it = s->graphs.find(key);
  if (it == s->graphs.end()) {
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    static std::mutex capture_mu;                      // captures are rare; keep them off each other's toes
    std::lock_guard<std::mutex> lk(capture_mu);
    WB_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    int rc = WB_OK;
    for (int i = 0; i < reps && rc == WB_OK; i++) rc = enqueue_step(s, n_launch, k, use_mask, fuse_ln, max_nb, false, chained, eot, bio);
    hipError_t e = hipStreamEndCapture(st, &g);
    WB_TRY(rc);
    WB_HIP(e);
    WB_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    it = s->graphs.emplace(key, ge).first;
  }
  WB_HIP(hipGraphLaunch(it->second, st));
  return WB_OK;
}

}  // extern "C"

namespace wb {
// The whole chained greedy decode as ONE persistent launch (decode_persist.hip): every sublayer of every step runs in a
// co-resident grid whose blocks hand their output planes to each other through arrival counters.  Enqueues the first
// step's prepare kernel, the control block and the launch, then waits for the stream.  *steps_done = steps executed.
//
// *fell_back: the launch was refused (no cooperative launch on this device / partition, the grid not co-resident, a
// second cooperative client) or a wait gave up before ANY step was committed -- the caller re-seeds the control block and
// runs the graph-replayed chain of one launch per sublayer instead (it derives everything from gctl), and the session
// stops trying the persistent kernel.  A wait that gives up after steps were committed stays an error.
static int run_persistent_chain(wb_session* s, int eot, int max_depth, int mask_until_len, const int32_t* forced, int n_forced,
                                int* steps_done, bool* fell_back) {
  *fell_back = false;
  WB_REQUIRE(n_forced >= 0 && n_forced <= PS_MAX_FORCED, WB_ERR_ARG, "persistent decode: %d prompt steps", n_forced);
  // test hook (tests/test_emu_functional.py, tests/test_gpu_switches.py): "launch" = behave as if the cooperative launch was refused
  static const char* inject = getenv("WHISPER_HIP_PERSIST_INJECT_FAIL");
  wb_model* m = s->m;
  const wb_dims& D = m->dims;
  const int d = D.n_text_state, H = D.n_text_head, NL = D.n_text_layer, V = D.n_vocab, S = s->S, W = s->W;
  const StepLayout& L = s->lay;
  hipStream_t st = s->st;
  const int NB = dec_mlp_fused_planes(d);
  const int n_tiles = (V + 127) / 128;
  const size_t pool = (size_t)s->Lmax * S;
  const int ldkv = 2 * d;
  // ---- the hand-off buffers: residual streams and partial planes as {tag, value} granules (zero-filled once: tag 0 is
  // never used; the tags of a launch live above launch_count << 16, so leftovers of earlier decodes never match)
  WB_REQUIRE(NB <= 32 && H <= 8, WB_ERR_SHAPE, "persistent decode: more planes than its folds hold");
  WB_TRY(s->ps_gx.ensure_zeroed(((size_t)2 * S + 8) * d * 8, st));
  WB_TRY(s->ps_gpa.ensure_zeroed(((size_t)H * S + 8) * d * 8, st));
  WB_TRY(s->ps_gpc.ensure_zeroed(((size_t)H * S + 8) * d * 8, st));
  WB_TRY(s->ps_gp2.ensure_zeroed(((size_t)NB * S + 8) * d * 8, st));
  WB_TRY(s->ps_gxn.ensure_zeroed(((size_t)S + 8) * d * 8, st));
  if (((s->ps_launches + 1) & 0xffffu) == 0) {     // the 16-bit launch count wraps: forget every old tag
    for (DevMem* b : {&s->ps_gx, &s->ps_gpa, &s->ps_gpc, &s->ps_gp2, &s->ps_gxn}) WB_HIP(hipMemsetAsync(b->p, 0, b->bytes, st));
    s->ps_launches++;
  }
  const unsigned tag_base = ((++s->ps_launches) & 0xffffu) << 16;
  char* gxb[2] = {static_cast<char*>(s->ps_gx.p), static_cast<char*>(s->ps_gx.p) + (size_t)S * d * 8};
  // ---- per-layer arguments: exactly what enqueue_step hands the fused sublayer kernels ----
  std::vector<PsLayerArgs> la(NL);
  float* xb[2] = {s->x.as<float>(), s->x.as<float>() + (size_t)S * d};
  int xi = 0;
  for (int l = 0; l < NL; l++) {
    const DecBlockW& b = m->dec[l];
    AttnFusedArgs& fa = la[l].attn;
    fa.st = s->state.as<int>(); fa.lay = L; fa.S = S; fa.d = d; fa.n_head = H;
    fa.x_in = xb[xi]; fa.pend = l == 0 ? nullptr : s->P2.as<float>(); fa.KSp = l == 0 ? 0 : NB;
    fa.pbias = l == 0 ? nullptr : m->dec[l - 1].mlp2.b; fa.x_out = xb[xi ^ 1];
    fa.ln_g = b.ln1.g; fa.ln_b = b.ln1.b; fa.ln_eps = b.ln1.eps; fa.ln_inside = m->ln_eps_inside_sqrt;
    fa.Wqkv = b.qkv.w; fa.ldqkv = b.qkv.n; fa.bqkv = b.qkv.b; fa.scale = m->qk_scale;
    fa.Kc = s->kc.as<float>() + (size_t)l * pool * d; fa.Vc = s->vc.as<float>() + (size_t)l * pool * d;
    fa.tabs = s->tabs.as<int>(); fa.Lmax = s->Lmax; fa.Wo = b.out.w; fa.P = s->Pa.as<float>();
    fa.g_x_in = gxb[xi]; fa.g_pend = l == 0 ? nullptr : s->ps_gp2.p; fa.g_x_out = gxb[xi ^ 1]; fa.g_P = s->ps_gpa.p;
    xi ^= 1;
    CrossFusedArgs& ca = la[l].cross;
    ca.st = s->state.as<int>(); ca.lay = L; ca.S = S; ca.d = d; ca.n_head = H;
    ca.x_in = xb[xi]; ca.pend = s->Pa.as<float>(); ca.KSp = H; ca.pbias = b.out.b; ca.x_out = xb[xi ^ 1];
    ca.ln_g = b.ln2.g; ca.ln_b = b.ln2.b; ca.ln_eps = b.ln2.eps; ca.ln_inside = m->ln_eps_inside_sqrt;
    ca.Wq = b.cq.w; ca.bq = b.cq.b; ca.scale = m->qk_scale;
    ca.ckv = s->ckv.as<float>() + (size_t)l * s->enc_rows * 2 * d; ca.ldkv = ldkv; ca.koff = 0;   // layer-major cached K|V
    ca.win_row0 = s->win_meta.as<int>(); ca.win_C = s->win_meta.as<int>() + W;
    ca.Wo = b.cout.w; ca.P = s->Pc.as<float>();
    ca.n_pass = s->maxC > CROSS_FUSED_MAX_C ? 2 : 1;
    ca.g_x_in = gxb[xi]; ca.g_pend = s->ps_gpa.p; ca.g_x_out = gxb[xi ^ 1]; ca.g_P = s->ps_gpc.p;
    xi ^= 1;
    MlpFusedArgs& ma = la[l].mlp;
    ma.st = s->state.as<int>(); ma.S = S; ma.d = d;
    ma.x_in = xb[xi]; ma.pend = s->Pc.as<float>(); ma.KSp = H; ma.pbias = b.cout.b; ma.x_out = xb[xi ^ 1];
    ma.ln_g = b.ln3.g; ma.ln_b = b.ln3.b; ma.ln_eps = b.ln3.eps; ma.ln_inside = m->ln_eps_inside_sqrt;
    ma.W1 = b.mlp1.w; ma.ld1 = b.mlp1.n; ma.b1 = b.mlp1.b; ma.W2 = b.mlp2.w; ma.P = s->P2.as<float>();
    ma.g_x_in = gxb[xi]; ma.g_pend = s->ps_gpc.p; ma.g_x_out = gxb[xi ^ 1]; ma.g_P = s->ps_gp2.p;
    xi ^= 1;
  }
  // ---- one step's roles, dealt to the blocks: every block runs its own list, in dependency order, every step ----
  std::vector<PsRole> lr;                            // the layer roles in dependency order
  for (int l = 0; l < NL; l++) {
    for (int r = 0; r < W; r++) for (int h = 0; h < H; h++) lr.push_back(PsRole{PSR_ATTN, l, h, r});
    for (int r = 0; r < W; r++) for (int h = 0; h < H; h++) lr.push_back(PsRole{PSR_CROSS, l, h, r});
    for (int j = 0; j < NB; j++) lr.push_back(PsRole{PSR_MLP, l, j, 0});
  }
  const int grid = std::max(1, std::min(s->ps_grid, (int)lr.size() + n_tiles + 2 * W));
  std::vector<std::vector<PsRole>> deal(grid);
  for (size_t i = 0; i < lr.size(); i++) deal[i % grid].push_back(lr[i]);
  // logits: blocks that hold a first-layer attention role stay free of it -- they are the first to be needed in the next
  // step and should be back at their wait (weights requested) before this one ends.  Every logits block takes a run of
  // consecutive 128-column tiles behind ONE fold + LayerNorm.
  std::vector<int> cand;
  for (int b = 0; b < grid; b++) {
    bool early = false;
    for (const PsRole& r : deal[b]) early |= r.layer == 0 && (r.kind == PSR_ATTN || r.kind == PSR_CROSS);
    if (!early) cand.push_back(b);
  }
  if ((int)cand.size() * 4 < n_tiles) { cand.clear(); for (int b = 0; b < grid; b++) cand.push_back(b); }
  // final LayerNorm (one per row): blocks without any layer role if there are some (they sit between the last MLP and
  // the logits on the critical path), else the least loaded ones; they take no logits work
  std::vector<char> is_fin(grid, 0);
  {
    std::vector<int> order(grid);
    for (int b = 0; b < grid; b++) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return deal[x].size() < deal[y].size(); });
    for (int r = 0; r < W; r++) { deal[order[r % grid]].push_back(PsRole{PSR_FINLN, 0, 0, r}); is_fin[order[r % grid]] = 1; }
  }
  if ((int)cand.size() > 2 * W) cand.erase(std::remove_if(cand.begin(), cand.end(), [&](int b) { return is_fin[b] != 0; }), cand.end());
  std::stable_sort(cand.begin(), cand.end(), [&](int x, int y) { return deal[x].size() < deal[y].size(); });
  const int tpb = (n_tiles + (int)cand.size() - 1) / (int)cand.size();
  const int n_lg = (n_tiles + tpb - 1) / tpb;
  for (int q = 0; q < n_lg; q++)
    deal[cand[q]].push_back(PsRole{PSR_LOGITS, q, q * tpb, std::min(tpb, n_tiles - q * tpb)});
  // merge (one per row): the least loaded blocks
  {
    std::vector<int> order(grid);
    for (int b = 0; b < grid; b++) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return deal[x].size() < deal[y].size(); });
    for (int r = 0; r < W; r++) deal[order[r % grid]].push_back(PsRole{PSR_MERGE, 0, 0, r});
  }
  std::vector<PsRole> roles;
  std::vector<int> role_off(grid + 1, 0);
  for (int b = 0; b < grid; b++) {
    role_off[b] = (int)roles.size();
    roles.insert(roles.end(), deal[b].begin(), deal[b].end());
  }
  role_off[grid] = (int)roles.size();
  const int n_ctl = ps_ctl_ints(S, NL);
  WB_TRY(s->ps_layers.ensure(la.size() * sizeof(PsLayerArgs)));
  WB_TRY(s->ps_roles.ensure(roles.size() * sizeof(PsRole) + role_off.size() * 4));
  WB_TRY(s->ps_ctl.ensure((size_t)n_ctl * 4));
  WB_TRY(s->ps_dead.ensure((size_t)S * 4));
  WB_TRY(s->ps_tstats.ensure((size_t)S * n_tiles * 2 * 4));
  std::vector<int> ctl0(n_ctl, 0);
  ctl0[HX_STOP] = INT_MAX;
  WB_HIP(hipMemcpyAsync(s->ps_layers.p, la.data(), la.size() * sizeof(PsLayerArgs), hipMemcpyHostToDevice, st));
  WB_HIP(hipMemcpyAsync(s->ps_roles.p, roles.data(), roles.size() * sizeof(PsRole), hipMemcpyHostToDevice, st));
  WB_HIP(hipMemcpyAsync(static_cast<char*>(s->ps_roles.p) + roles.size() * sizeof(PsRole), role_off.data(), role_off.size() * 4,
                        hipMemcpyHostToDevice, st));
  WB_HIP(hipMemcpyAsync(s->ps_ctl.p, ctl0.data(), (size_t)n_ctl * 4, hipMemcpyHostToDevice, st));
  WB_HIP(hipMemsetAsync(s->ps_dead.p, 0, (size_t)S * 4, st));
  PersistArgs a;
  a.layers = s->ps_layers.as<PsLayerArgs>(); a.roles = s->ps_roles.as<PsRole>(); a.n_roles = (int)roles.size();
  a.role_off = reinterpret_cast<const int*>(static_cast<char*>(s->ps_roles.p) + roles.size() * sizeof(PsRole));
  a.n_logits_roles = n_lg;
  a.n_layer = NL; a.n_rows = W; a.S = S; a.d = d; a.n_head = H; a.nb_mlp = NB;
  a.n_pass = s->maxC > CROSS_FUSED_MAX_C ? 2 : 1;
  a.ctl = s->ps_ctl.as<int>(); a.step0 = s->step; a.n_steps = n_forced + max_depth; a.mask_until_len = mask_until_len;
  a.n_forced = n_forced;
  for (int i = 0; i < n_forced; i++) a.forced[i] = forced[i];
  a.g_xn = s->ps_gxn.p;
  a.x_fin = gxb[xi]; a.P2 = s->ps_gp2.p; a.b2_last = m->dec[NL - 1].mlp2.b; a.tag_base = tag_base;
  a.ln_g = m->ln_dec.g; a.ln_b = m->ln_dec.b; a.ln_eps = m->ln_dec.eps; a.ln_inside = m->ln_eps_inside_sqrt;
  a.Et = m->tok_emb_t; a.vocab_ld = m->vocab_ld; a.V = V; a.mask = s->mask.as<float>();
  a.tstats = s->ps_tstats.as<float>(); a.n_tiles = n_tiles;
  a.gctl = s->gctl.as<int>(); a.gtok = s->gtok.as<int>(); a.Lmax = s->Lmax; a.eot = eot;
  a.E = m->tok_emb; a.pos = m->dec_pos; a.x0 = gxb[0]; a.tabs = s->tabs.as<int>(); a.dead = s->ps_dead.as<int>();
  // optional role timeline (developer): WHISPER_HIP_PS_STAMPS=<file> dumps [n_steps][n_roles][3] 100 MHz clock values
  static const char* stamps_path = getenv("WHISPER_HIP_PS_STAMPS");
  const size_t n_stamps = stamps_path ? (size_t)(n_forced + max_depth) * roles.size() * 8 : 0;
  if (n_stamps) {
    WB_TRY(s->ps_stamps.ensure(n_stamps * 8));
    WB_HIP(hipMemsetAsync(s->ps_stamps.p, 0, n_stamps * 8, st));
    a.stamps = s->ps_stamps.as<unsigned long long>();
  }
  WB_REQUIRE(3 * NL * (n_forced + max_depth + 1) + 4 < 0x10000, WB_ERR_SHAPE,
             "persistent decode: %d layers x %d steps do not fit the 16-bit granule tags", NL, max_depth);
  // first step of the chain: token + position embedding of the last prompt token (every later step: the merge role)
  launch_dec_prepare(st, reinterpret_cast<const int*>(s->host_block_dev), s->state.as<int>(), L, W, s->tabs.as<int>(),
                     s->Lmax, m->tok_emb, m->dec_pos, d, s->x.as<float>(), s->gctl.as<int>());
  launch_ps_seed(st, s->x.as<float>(), W * d, gxb[0], tag_base + 1u);
  WB_HIP(hipStreamSynchronize(st));            // (the staging vectors above are on the stack)
  hipEvent_t e0 = nullptr, e1 = nullptr;
  prof_tag(KC_PERSIST, 0.0);                   // (its necessary bytes are known when the rows' lengths are: added by the caller)
  const bool timed = prof_take_events(&e0, &e1);
  if (timed) WB_HIP(hipEventRecord(e0, st));
  const bool refused = (inject && !strcmp(inject, "launch")) || launch_dec_persist(st, a, grid) != 0;
  if (timed) WB_HIP(hipEventRecord(e1, st));
  if (refused) {
    (void)hipGetLastError();                   // (clears the sticky launch error)
    s->ps_grid = 0;
    *fell_back = true;
    *steps_done = 0;
    return WB_OK;
  }
  std::vector<int> ctl(n_ctl);
  WB_HIP(hipMemcpyAsync(ctl.data(), s->ps_ctl.p, (size_t)n_ctl * 4, hipMemcpyDeviceToHost, st));
  int gstep = 0;
  WB_HIP(hipMemcpyAsync(&gstep, s->gctl.as<int>() + GC_STEP, 4, hipMemcpyDeviceToHost, st));
  WB_HIP(hipStreamSynchronize(st));
  if (ctl[HX_ERR] != 0 && gstep == s->step) {  // nothing committed: the chain can take over from the same control block
    s->ps_grid = 0;
    *fell_back = true;
    *steps_done = 0;
    return WB_OK;
  }
  WB_REQUIRE(ctl[HX_ERR] == 0, WB_ERR_HIP, "persistent decode: a wait gave up (counter %d) at step %d", ctl[HX_ERR] - 1, gstep);
  if (n_stamps) {
    std::vector<unsigned long long> hs(n_stamps);
    WB_HIP(hipMemcpy(hs.data(), s->ps_stamps.p, n_stamps * 8, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(stamps_path, "wb")) {
      const int hdr[4] = {n_forced + max_depth, (int)roles.size(), grid, 8};
      fwrite(hdr, 4, 4, f);
      std::vector<int> kinds(roles.size());
      for (size_t i = 0; i < roles.size(); i++)
        kinds[i] = roles[i].kind | ((roles[i].kind <= PSR_MLP ? roles[i].layer : 0) << 8) |
                   ((roles[i].kind == PSR_LOGITS ? 0 : roles[i].b) << 16);
      fwrite(kinds.data(), 4, kinds.size(), f);
      fwrite(hs.data(), 8, hs.size(), f);
      fclose(f);
    }
  }
  *steps_done = std::max(0, std::min(n_forced + max_depth, gstep - s->step));
  return WB_OK;
}

// Device-chained greedy decode (beam_size == 1): after the host-driven prompt prefill, every step's
// argmax is fed to the next step on the device; the host only replays the step graph and checks the
// per-window finished flags every `chunk` steps.  Equivalent to beam.rs with k = 1: the single beam is
// extended by its best continuation (lowest id on ties) until it ends in EOT or max_depth tokens.
// A fresh session (step 0) hands in the whole prompt: the persistent kernel runs the prompt's first prompt_len - 1 positions as
// forced steps of the same launch (no host round trip between the prompt and the first generated token); the chain of one
// launch per sublayer (and the persistent kernel's fallback) prefills through host-driven steps first, as before.
// The result of a device-chained beam search from its control block (copied to the host): per window the sequence of the
// max-log-prob beam -- beam.rs:33-36, the LAST of the equal maxima -- walked back through the node tree.
static int beam_chain_extract(const std::vector<int>& ctl, const BeamChainLayout& bl, const int32_t* prompt, int P,
                              int32_t* out_tokens, int32_t row_stride, int32_t* out_lens, bool* all_done_out) {
  const double* lp = reinterpret_cast<const double*>(ctl.data() + bl.lp);
  const int* nodes = ctl.data() + bl.nodes;
  bool all_done = true;
  for (int w = 0; w < bl.W; w++) {
    const int nb = ctl[bl.nb + w];
    int best = -1;
    for (int i = 0; i < nb; i++) {
      WB_REQUIRE(lp[w * BEAM_KB + i] == lp[w * BEAM_KB + i], WB_ERR_STATE, "beam search: NaN log-probability (reference panics)");
      if (best < 0 || lp[w * BEAM_KB + i] >= lp[w * BEAM_KB + best]) best = i;
    }
    std::vector<int32_t> seq;
    for (int nd = best >= 0 ? ctl[bl.node + w * BEAM_KB + best] : -1; nd >= 0; nd = nodes[2 * nd + 1]) seq.push_back(nodes[2 * nd]);
    std::reverse(seq.begin(), seq.end());
    const int len = (P - 1) + (int)seq.size();
    WB_REQUIRE(len <= row_stride, WB_ERR_ARG, "row_stride too small");
    int32_t* row = out_tokens + (size_t)w * row_stride;
    for (int i = 0; i < P - 1; i++) row[i] = prompt[i];
    for (size_t i = 0; i < seq.size(); i++) row[P - 1 + i] = seq[i];
    out_lens[w] = len;
    all_done = all_done && (ctl[bl.done + w] != 0 || (best >= 0 && ctl[bl.fin + w * BEAM_KB + best]));
  }
  *all_done_out = all_done;
  return WB_OK;
}

// The initial control block: one beam per window holding the prompt's last token (the rest of the prompt is the KV prefill).
static void beam_chain_init(std::vector<int>& ctl, const BeamChainLayout& bl, const int32_t* prompt, int P, int eot) {
  ctl.assign((size_t)bl.total_ints, 0);
  double* lp = reinterpret_cast<double*>(ctl.data() + bl.lp);
  int* nodes = ctl.data() + bl.nodes;
  for (int w = 0; w < bl.W; w++) {
    const int nd = w * BEAM_KB;                   // level 0 of the pool
    ctl[bl.nb + w] = 1;
    ctl[bl.node + w * BEAM_KB] = nd;
    nodes[2 * nd] = prompt[P - 1]; nodes[2 * nd + 1] = -1;
    ctl[bl.fin + w * BEAM_KB] = prompt[P - 1] == eot ? 1 : 0;     // transcribe.rs:235-241
    ctl[bl.prev_slot + w * BEAM_KB] = P > 1 ? w : -1;
    lp[w * BEAM_KB] = 0.0;
  }
}

// Beam search with the bookkeeping on the device.  What the host still does: the prompt prefill (P - 1 ordinary steps), the
// initial control block, enqueueing the steps (whole chunks as one graph launch), one synchronisation per chunk to see whether
// every window has ended, and the walk back through the node tree at the end.  Results are those of beam_search_windows
// (transcribe.cpp) step for step: same top-k rows, same f64 sums, same insertion and tie rules.
int session_beam_chain(wb_session* s, const int32_t* prompt, int P, int k, int eot, int max_depth, int mask_until_len,
                       int32_t* out_tokens, int32_t row_stride, int32_t* out_lens, bool* handled) {
  *handled = false;
  wb_model* m = s->m;
  const wb_dims& D = m->dims;
  const int W = s->W, S = s->S, V = D.n_vocab;
  static const bool enabled = []() { const char* e = getenv("WHISPER_HIP_BEAM_CHAIN"); return !(e && e[0] == '0'); }();
  if (!enabled || W > 64 || k < 1 || k > TOPK_MAX || k > s->max_beams || max_depth <= 0 || s->step != 0 || P < 1 || W * k > S)
    return WB_OK;
  WB_REQUIRE(s->has_mask || mask_until_len < P, WB_ERR_STATE, "wb_session_decode: special mask not set");
  *handled = true;
  const int asked_depth = max_depth;
  max_depth = std::min(max_depth, s->Lmax - (P - 1));
  WB_HIP(hipSetDevice(m->device));
  hipStream_t st = s->st;
  const StepLayout& L = s->lay;
  {  // prefill: all prompt tokens but the last only feed the KV cache (beam_search_windows does the same)
    std::vector<int32_t> tok(W), par(W), win(W);
    for (int t = 0; t < P - 1; t++) {
      for (int w = 0; w < W; w++) { tok[w] = prompt[t]; par[w] = t == 0 ? -1 : w; win[w] = w; }
      WB_TRY(wb_session_step(s, tok.data(), par.data(), win.data(), W, 0, 0, nullptr, nullptr));
    }
  }
  const BeamChainLayout bl = make_beam_layout(W, std::max(max_depth, 0));
  WB_TRY(s->bc_ctl.ensure((size_t)bl.total_ints * 4));
  WB_TRY(s->bc_topk.ensure((size_t)S * TOPK_MAX * 8));
  std::vector<int> ctl;
  beam_chain_init(ctl, bl, prompt, P, eot);
  WB_HIP(hipMemcpyAsync(s->bc_ctl.p, ctl.data(), ctl.size() * 4, hipMemcpyHostToDevice, st));
  WB_HIP(hipStreamSynchronize(st));
  // launch shape: the bucket wb_session_step would pick for the most rows the search can have live (W k)
  const int n = W * k;
  static const bool fuse16_enabled = []() {
    const char* e = getenv("WHISPER_HIP_FUSE16"); const char* f = getenv("WHISPER_HIP_FUSE_SUB"); const char* x = getenv("WHISPER_HIP_FUSE_X");
    const char* c = getenv("WHISPER_HIP_FUSE_CO");
    return !(e && e[0] == '0') && !(f && f[0] == '0') && !(x && x[0] == '0') && !(c && c[0] == '1');
  }();
  const bool g16 = fuse16_enabled && n > 8 && n <= 16 && S >= 9 && dec_fused_supported(D.n_text_state) &&
                   D.n_text_state == 64 * D.n_text_head && s->maxC <= CROSS_FUSED_MAX_PASSES * CROSS_FUSED_MAX_C;
  const int n_launch = n <= 4 ? std::min(4, S) : n <= 8 ? std::min(8, S) : g16 ? std::min(16, S) : S;
  const bool fuse_ln = n_launch <= 8 || g16;
  const int max_nb = s->max_beams <= 1 ? 1 : s->max_beams <= 2 ? 2 : s->max_beams <= 4 ? 4 : 8;
  static const bool graphs_enabled = []() { const char* e = getenv("WHISPER_HIP_GRAPH"); return !(e && e[0] == '0'); }();
  const bool use_graph = graphs_enabled && !profile().on;
  BeamStepIO bio;
  bio.state_src = s->state.as<int>();
  bio.topk_id = s->bc_topk.as<int32_t>();
  bio.topk_lp = reinterpret_cast<float*>(s->bc_topk.as<int32_t>() + (size_t)S * TOPK_MAX);
  bio.upd.ctl = s->bc_ctl.as<int>(); bio.upd.bl = bl; bio.upd.topk_id = bio.topk_id; bio.upd.topk_lp = bio.topk_lp;
  // the bookkeeping kernel writes the device state block the step kernels read, and prepares the step's rows itself
  bio.upd.state_out = s->state.as<int>(); bio.upd.lay = L;
  bio.upd.tabs = s->tabs.as<int>(); bio.upd.Lmax = s->Lmax; bio.upd.E = m->tok_emb; bio.upd.pos = m->dec_pos;
  bio.upd.d = D.n_text_state; bio.upd.x = s->x.as<float>(); bio.upd.k = k; bio.upd.eot = eot; bio.upd.V = V;
  bio.upd.first = 0; bio.upd.step_pos = P - 1;
  // the captured graphs bake in the search's constants: drop them when those differ from the last search of this session
  const uint64_t bsig = ((uint64_t)max_depth << 40) ^ ((uint64_t)(P - 1) << 24) ^ ((uint64_t)k << 16) ^ (uint64_t)(unsigned)eot;
  if (bsig != s->beam_sig) { s->clear_graphs(); s->beam_sig = bsig; }
  ScopedTimer tm(st, 3);
  {
    BeamChainArgs a0 = bio.upd;
    a0.first = 1;                                  // termination test + slots of the first step (beam.rs:23-27 runs BEFORE the step)
    launch_dec_beam_update(st, a0);
  }
  const int chunk = 16;
  int depth = 0;
  int hdr[BC_HDR] = {0};
  while (depth < max_depth) {
    int enq = 0;
    while (depth + enq < max_depth && enq < chunk) {
      const int d0 = depth + enq;
      const int use_mask = (P + d0) <= mask_until_len ? 1 : 0;     // transcribe.rs:271-275
      const int run = (!use_mask && max_depth - d0 >= chunk && enq == 0) ? chunk : 1;
      WB_TRY(launch_step(s, n_launch, k, use_mask, fuse_ln, max_nb, use_graph, false, eot, run, &bio));
      if (profile().on) profile().ms[4] += run;
      enq += run;
    }
    depth += enq;
    WB_HIP(hipMemcpyAsync(hdr, s->bc_ctl.p, sizeof(hdr), hipMemcpyDeviceToHost, st));
    WB_HIP(hipStreamSynchronize(st));
    if (hdr[BC_ALLDONE] || hdr[BC_ERR]) break;     // every window has ended: the kernels of further steps would exit at once
  }
  tm.stop();
  WB_HIP(hipMemcpyAsync(ctl.data(), s->bc_ctl.p, ctl.size() * 4, hipMemcpyDeviceToHost, st));
  WB_HIP(hipStreamSynchronize(st));
  tm.collect();
  if (profile().on) prof_collect();
  s->prof_step_off = 0;
  const int steps_done = ctl[BC_DEPTH];
  s->step += steps_done;
  s->prev_n = 0; s->prev_len.clear(); s->prev_win.clear();     // (the device-side slots are not mirrored: no host-driven step may follow)
  s->last_had_logits = 0;
  WB_TRY(dec_split_check(s));
  WB_REQUIRE(ctl[BC_ERR] == 0, WB_ERR_STATE, "beam search: NaN log-probability (reference panics)");
  {
    bool all_done = true;
    WB_TRY(beam_chain_extract(ctl, bl, prompt, P, out_tokens, row_stride, out_lens, &all_done));
    if (max_depth < asked_depth && !all_done)
      WB_REQUIRE(false, WB_ERR_SHAPE, "Token sequence length %d must not exceed %d.", s->Lmax + 1, s->Lmax);
  }
  return WB_OK;
}

int session_greedy_chain(wb_session* s, const int32_t* prompt, int eot, int max_depth, int mask_until_len, int prompt_len,
                         int32_t* out_tokens, int32_t row_stride, int32_t* out_lens) {
  wb_model* m = s->m;
  const int S = s->S, W = s->W;
  WB_REQUIRE(S == W, WB_ERR_STATE, "chained greedy decode needs max_beams == 1");
  WB_REQUIRE(prompt && prompt_len >= 1, WB_ERR_ARG, "chained greedy decode: empty prompt");
  WB_REQUIRE(s->step == 0 || (s->prev_n == W && s->step == prompt_len - 1), WB_ERR_STATE,
             "chained greedy decode: the session is neither fresh nor prefilled");
  const int n_forced_max = prompt_len - 1;          // prompt positions that can run inside the persistent launch
  auto host_prefill = [&]() -> int {                 // transcribe.rs:203: the prompt, one KV-cached step per token, no logits
    std::vector<int32_t> tok(W), par(W), win(W);
    for (int t = s->step; t < prompt_len - 1; t++) {
      for (int w = 0; w < W; w++) { tok[w] = prompt[t]; par[w] = t == 0 ? -1 : w; win[w] = w; }
      WB_TRY(wb_session_step(s, tok.data(), par.data(), win.data(), W, 0, 0, nullptr, nullptr));
    }
    return WB_OK;
  };
  // the first steps read the special-token mask (transcribe.rs:271-275): same contract as wb_session_step
  WB_REQUIRE(s->has_mask || mask_until_len < prompt_len || max_depth == 0, WB_ERR_STATE,
             "wb_session_decode: special mask not set");
  // The reference only fails (mod.rs:134-139) when a sequence actually outgrows n_text_ctx: a large max_depth
  // whose windows all end on EOT earlier succeeds.  Run at most the steps the context holds; raise the
  // reference's error afterwards if a window is still unfinished.
  const int asked_depth = max_depth;
  max_depth = std::min(max_depth, s->Lmax - (prompt_len - 1));
  WB_HIP(hipSetDevice(m->device));
  hipStream_t st = s->st;
  const size_t ctl_ints = GC_HDR + 3 * (size_t)S;
  WB_TRY(s->gctl.ensure(ctl_ints * 4));
  // token rows [S][Lmax]; the last generated token of a row that fills the context lands at index Lmax
  // (= slot 0 of the next row, a prompt position nobody reads), so the buffer carries one extra slot
  WB_TRY(s->gtok.ensure(((size_t)S * s->Lmax + 1) * 4));
  std::vector<int> ctl(ctl_ints, 0);
  auto seed_ctl = [&]() -> int {                     // the chain starts at the session's step with that position's prompt token
    std::fill(ctl.begin(), ctl.end(), 0);
    ctl[GC_STEP] = s->step;
    for (int i = 0; i < W; i++) ctl[GC_HDR + i] = prompt[s->step];
    WB_HIP(hipMemcpyAsync(s->gctl.p, ctl.data(), ctl_ints * 4, hipMemcpyHostToDevice, st));
    WB_HIP(hipStreamSynchronize(st));
    return WB_OK;
  };
  const int n_launch = W <= 4 ? std::min(4, S) : W <= 8 ? std::min(8, S) : S;
  const bool fuse_ln = n_launch <= 8;
  static const bool graphs_enabled = []() { const char* e = getenv("WHISPER_HIP_GRAPH"); return !(e && e[0] == '0'); }();
  const bool use_graph = graphs_enabled && !profile().on;
  const int chunk = 16;
  int depth = 0;                                   // steps enqueued so far
  // the persistent flag-chained kernel (decode_persist.hip): the small-batch fused path of exact-f32 models whose roles
  // fit one co-resident grid.  Default since it passed the graph-replayed chain of one launch per sublayer on MI355X
  // (14.76 vs 15.5 ms per 30 s of tiny.en audio, profiles/r03_c_ab_*); WHISPER_HIP_PERSIST=0 selects the chain.
  static const bool persist_enabled = []() { const char* e = getenv("WHISPER_HIP_PERSIST"); return !(e && e[0] == '0'); }();
  static const bool fused_enabled = []() {
    const char* a = getenv("WHISPER_HIP_FUSE_SUB"); const char* b = getenv("WHISPER_HIP_FUSE_X");
    return !(a && a[0] == '0') && !(b && b[0] == '0');
  }();
  bool persist = persist_enabled && fused_enabled && fuse_ln && max_depth > 0 &&
                 dec_fused_supported(m->dims.n_text_state) && m->dims.n_text_state == 64 * m->dims.n_text_head &&
                 dec_persist_supported(m->dims.n_text_state, W, s->maxC);
  if (persist) {
    if (s->ps_grid < 0) s->ps_grid = dec_persist_max_grid(m->device, m->dims.n_text_state, W, s->maxC);
    persist = s->ps_grid > 0;
  }
  // prompt positions the persistent launch runs itself (a fresh session; WHISPER_HIP_PERSIST_PREFILL=0: host-driven prefill)
  static const bool fold_prefill = []() { const char* e = getenv("WHISPER_HIP_PERSIST_PREFILL"); return !(e && e[0] == '0'); }();
  int n_forced = 0;
  if (persist && fold_prefill && s->step == 0 && n_forced_max <= PS_MAX_FORCED && s->has_mask) n_forced = n_forced_max;
  if (n_forced == 0) WB_TRY(host_prefill());
  WB_TRY(seed_ctl());
  ScopedTimer tm(st, 3);
  if (persist) {
    bool fell_back = false;
    WB_TRY(run_persistent_chain(s, eot, max_depth, mask_until_len, prompt + s->step + 1, n_forced, &depth, &fell_back));
    if (fell_back) {
      // rows whose merge role ran before the give-up have moved their control words: prefill on the host (if the launch was
      // to do it) and start the chain over from the seed
      persist = false;
      depth = 0;
      WB_TRY(host_prefill());
      WB_TRY(seed_ctl());
      n_forced = 0;
    } else {
      if (profile().on) profile().ms[4] += depth;
      depth -= n_forced;                             // from here on `depth` counts GENERATED positions
      s->step += n_forced;
      s->prev_n = W;
      s->prev_win.resize(W);
      for (int w = 0; w < W; w++) s->prev_win[w] = w;
    }
  }
  if (!persist) {
  if (fuse_ln)   // first step of the chain; every later one is prepared by its predecessor's merge kernel
    launch_dec_prepare(st, reinterpret_cast<const int*>(s->host_block_dev), s->state.as<int>(), s->lay, n_launch,
                       s->tabs.as<int>(), s->Lmax, m->tok_emb, m->dec_pos, m->dims.n_text_state, s->x.as<float>(),
                       s->gctl.as<int>());
  // masked steps (the first two, transcribe.rs:271-275) and the tail shorter than a chunk go one step per
  // graph launch; in between, a whole chunk of steps is ONE graph launch (two multi-step shapes are never
  // needed: only {1 step masked, 1 step, chunk steps} are captured).
  //
  // Small batches (the fused-LayerNorm path) run one segment AHEAD of the finished flags: segment k + 1 is enqueued
  // before the flags of segment k are read (on a second stream, behind an event), so the GPU never idles across
  // the host round trip (~60-400 us per check in round 1's timeline).  If every window turns out to be finished,
  // the merge kernel has already blanked the step state (ST_N = 0) and the kernels of the speculative segment exit
  // at their first instruction.  Batch mode (> 8 rows: MFMA GEMMs that do not look at ST_N) keeps the blocking check.
  // (opt-in: measured 1417x vs 1543x -- a graph launched behind a running graph starts later than one launched on an
  // idle stream saves)
  static const bool spec_enabled = []() { const char* e = getenv("WHISPER_HIP_SPECULATE"); return e && e[0] == '1'; }();
  const bool speculate = fuse_ln && spec_enabled;
  if (speculate && !s->st2) {
    WB_HIP(hipStreamCreateWithFlags(&s->st2, hipStreamNonBlocking));
    WB_HIP(hipEventCreateWithFlags(&s->ev_seg, hipEventDisableTiming));
  }
  auto enqueue_segment = [&](int* enq) -> int {   // >= `chunk` steps (or what is left); returns steps enqueued via *enq
    int n = 0;
    while (depth + n < max_depth && n < chunk) {
      const int d0 = depth + n;
      const int use_mask = (prompt_len + d0) <= mask_until_len ? 1 : 0;
      const int run = (!use_mask && max_depth - d0 >= chunk) ? chunk : 1;
      WB_TRY(launch_step(s, n_launch, 1, use_mask, fuse_ln, 1, use_graph, true, eot, run));
      if (profile().on) profile().ms[4] += run;
      n += run;
    }
    *enq = n;
    return WB_OK;
  };
  auto read_flags = [&](bool behind_event) -> int {
    if (behind_event) {
      WB_HIP(hipStreamWaitEvent(s->st2, s->ev_seg, 0));
      WB_HIP(hipMemcpyAsync(ctl.data(), s->gctl.p, ctl_ints * 4, hipMemcpyDeviceToHost, s->st2));
      WB_HIP(hipStreamSynchronize(s->st2));
    } else {
      WB_HIP(hipMemcpyAsync(ctl.data(), s->gctl.p, ctl_ints * 4, hipMemcpyDeviceToHost, st));
      WB_HIP(hipStreamSynchronize(st));
    }
    return WB_OK;
  };
  auto all_done = [&]() {
    bool d = true;
    for (int i = 0; i < W; i++) d = d && ctl[GC_HDR + S + i] != 0;
    return d;
  };
  // small batches: the merge kernel publishes (steps completed, finished) per row into mapped host memory; the host
  // spins on it (a few us) instead of a D2H copy + stream synchronisation (30-190 us of idle GPU per check in round 1)
  static const bool poll_enabled = []() { const char* e = getenv("WHISPER_HIP_POLL"); return !(e && e[0] == '0'); }();
  const bool poll = fuse_ln && poll_enabled && !profile().on;
  volatile int* hfl = reinterpret_cast<volatile int*>(s->host_block + s->chain_flags_off);
  if (poll) for (int i = 0; i < 2 * S; i++) hfl[i] = 0;
  auto wait_flags = [&](int want_step, bool* done) -> int {
    for (long spins = 0;; spins++) {
      bool ready = true;
      for (int i = 0; i < W && ready; i++) ready = hfl[2 * i] >= want_step;
      if (ready) break;
      bool fin = true;
      for (int i = 0; i < W && fin; i++) fin = hfl[2 * i + 1] != 0;
      if (fin) break;                              // every window finished: the rest of the chunk is blanked
      if ((spins & 0xfff) == 0xfff) {             // every 4096 spins: is the stream still alive / busy?
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) {                     // stream drained: the flags are final (or the chain ended early)
          bool r2 = true;
          for (int i = 0; i < W && r2; i++) r2 = hfl[2 * i] >= want_step;
          if (r2) break;
          bool all = true;
          for (int i = 0; i < W; i++) all = all && hfl[2 * i + 1] != 0;
          if (all) break;                          // every window finished: later steps were blanked
          set_error("chained decode: the device stopped at step %d of %d", (int)hfl[0], want_step);
          return WB_ERR_HIP;
        }
        if (q != hipErrorNotReady) { set_error("chained decode: %s", hipGetErrorString(q)); return WB_ERR_HIP; }
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    bool all = true;
    for (int i = 0; i < W; i++) all = all && hfl[2 * i + 1] != 0;
    *done = all;
    return WB_OK;
  };
  if (!speculate) {
    while (depth < max_depth) {
      int n = 0;
      WB_TRY(enqueue_segment(&n));
      depth += n;
      if (poll) {
        bool done = false;
        WB_TRY(wait_flags(s->step + depth, &done));
        if (done) break;
      } else {
        WB_TRY(read_flags(false));
        if (all_done()) break;
      }
    }
  } else {
    int n_cur = 0;
    WB_TRY(enqueue_segment(&n_cur));
    depth += n_cur;
    while (true) {
      WB_HIP(hipEventRecord(s->ev_seg, st));        // end of the segment whose flags are read next
      int n_next = 0;
      if (depth < max_depth) { WB_TRY(enqueue_segment(&n_next)); depth += n_next; }
      WB_TRY(read_flags(true));
      if (all_done() || n_next == 0) break;
    }
    WB_HIP(hipStreamSynchronize(st));
  }
  }   // (!persist)
  tm.stop();
  std::vector<int> toks((size_t)S * s->Lmax + 1);
  WB_HIP(hipMemcpyAsync(ctl.data(), s->gctl.p, ctl_ints * 4, hipMemcpyDeviceToHost, st));
  WB_HIP(hipMemcpyAsync(toks.data(), s->gtok.p, toks.size() * 4, hipMemcpyDeviceToHost, st));
  WB_HIP(hipStreamSynchronize(st));
  tm.collect();
  WB_TRY(dec_split_check(s));
  WB_REQUIRE(ctl[GC_BAD] == 0, WB_ERR_STATE, "a decode step produced a row without a finite log-prob (NaN logits: non-finite "
             "weights or activations); its window was ended on <|endoftext|> on the device and the rows of this call are invalid");
  for (int w = 0; w < W; w++) {
    int len = ctl[GC_HDR + 2 * S + w];                 // prompt + generated (through EOT if it came)
    if (len < prompt_len) len = prompt_len;
    len = std::min(len, prompt_len + max_depth);
    WB_REQUIRE(len <= row_stride, WB_ERR_ARG, "row_stride too small");
    for (int p = prompt_len; p < len; p++) out_tokens[(size_t)w * row_stride + p] = toks[(size_t)w * s->Lmax + p];
    out_lens[w] = len;
  }
  if (profile().on && !persist && s->prof_cls_cross >= 0 && s->prof_cls_self >= 0) {
    // The tags counted every launched row's cached K/V.  A row whose window had already ended is marked dead in the
    // step state: its attention blocks exit at their first wait and stream nothing -- take those bytes back, so that
    // the reported algorithmic bytes are the NECESSARY ones.
    const int dm = m->dims.n_text_state, NL = m->dims.n_text_layer;
    double dead_ckv = 0, dead_self = 0;
    for (int w = 0; w < W; w++) {
      const int live = std::max(0, std::min(out_lens[w] - prompt_len, depth));   // steps in which row w was live
      for (int t = live; t < depth; t++) { dead_ckv += 8.0 * s->C[w] * dm; dead_self += 8.0 * (s->step + t + 1) * dm; }
    }
    prof_adjust_bytes(s->prof_cls_cross, -(double)NL * dead_ckv);
    prof_adjust_bytes(s->prof_cls_self, -(double)NL * dead_self);
  }
  if (profile().on && persist) {
    // necessary bytes of the persistent launch: weights + E^T once per executed step, a row's cached cross K/V and
    // self-attention rows only while its window is live
    const double dm = m->dims.n_text_state, NL = m->dims.n_text_layer;
    double bytes = (double)(depth + n_forced) * 4.0 * NL * 14.0 * dm * dm + (double)depth * 4.0 * (double)m->dims.n_vocab * dm;
    for (int w = 0; w < W; w++)                      // (the prompt positions the launch ran itself: every row is live there)
      for (int t = 0; t < n_forced; t++) bytes += NL * (8.0 * s->C[w] * dm + 8.0 * (t + 1) * dm);
    for (int w = 0; w < W; w++) {
      const int live = std::max(0, std::min(out_lens[w] - prompt_len, depth));
      for (int t = 0; t < live; t++) bytes += NL * (8.0 * s->C[w] * dm + 8.0 * (s->step + t + 1) * dm);
    }
    prof_adjust_bytes(KC_PERSIST, bytes);
  }
  s->prof_step_off = 0;
  s->step += depth;
  s->prev_len.assign(W, s->step);
  s->last_had_logits = 0;
  if (max_depth < asked_depth)
    for (int w = 0; w < W; w++)
      WB_REQUIRE(ctl[GC_HDR + S + w] != 0, WB_ERR_SHAPE, "Token sequence length %d must not exceed %d.", s->Lmax + 1,
                 s->Lmax);
  return WB_OK;
}
}  // namespace wb

// Test hook: beam.rs's bookkeeping as the DEVICE runs it (dec_beam_update_kernel), driven by a caller-supplied step function
// with wb_session_step's contract instead of the decoder -- the counterpart of wb_beam_search (the host restatement), so that
// the two can be compared on scripted log-prob rows with exact ties, finished beams and windows ending at different depths.
// No model: `device` only hosts the three small buffers.
extern "C" int wb_beam_search_device(int device, const wb_decode_params* p, int n_windows, int n_vocab, wb_step_fn step,
                                     void* user, int32_t* out_tokens, int32_t row_stride, int32_t* out_lens) {
  using namespace wb;
  WB_REQUIRE(p && step && out_tokens && out_lens && n_windows >= 1 && n_windows <= 64, WB_ERR_ARG, "wb_beam_search_device: bad argument");
  WB_REQUIRE(p->beam_size >= 1 && p->beam_size <= TOPK_MAX && p->max_depth >= 0, WB_ERR_ARG, "wb_beam_search_device: bad beam_size / max_depth");
  const int W = n_windows, k = p->beam_size, S = W * MAX_BEAMS, P = 4, V = n_vocab, eot = p->tok_end_of_text;
  const int32_t prompt[4] = {p->tok_start_of_transcript, p->tok_language, p->tok_transcribe, p->tok_no_timestamps};
  WB_REQUIRE(row_stride >= P + p->max_depth, WB_ERR_ARG, "row_stride %d < %d", row_stride, P + p->max_depth);
  wb::GpuTurn turn(device);
  WB_HIP(hipSetDevice(device));
  const StepLayout L = make_step_layout(S, W);
  const BeamChainLayout bl = make_beam_layout(W, p->max_depth);
  DevMem d_ctl, d_state, d_topk;
  WB_TRY(d_ctl.ensure((size_t)bl.total_ints * 4));
  WB_TRY(d_state.ensure((size_t)L.total * 4));
  WB_TRY(d_topk.ensure((size_t)S * TOPK_MAX * 8));
  std::vector<int32_t> tok(S), par(S), win(S);
  for (int t = 0; t < P - 1; t++) {               // the prompt prefill: steps without logits, as beam_search_windows issues them
    for (int w = 0; w < W; w++) { tok[w] = prompt[t]; par[w] = t == 0 ? -1 : w; win[w] = w; }
    WB_TRY(step(user, tok.data(), par.data(), win.data(), W, 0, 0, nullptr, nullptr));
  }
  std::vector<int> ctl;
  beam_chain_init(ctl, bl, prompt, P, eot);
  WB_HIP(hipMemcpy(d_ctl.p, ctl.data(), ctl.size() * 4, hipMemcpyHostToDevice));
  BeamChainArgs a;
  a.ctl = d_ctl.as<int>(); a.bl = bl; a.topk_id = d_topk.as<int32_t>();
  a.topk_lp = reinterpret_cast<float*>(d_topk.as<int32_t>() + (size_t)S * TOPK_MAX);
  a.state_out = d_state.as<int>(); a.lay = L; a.k = k; a.eot = eot; a.V = V; a.step_pos = P - 1;
  a.first = 1;
  launch_dec_beam_update(nullptr, a);
  a.first = 0;
  std::vector<int> st(L.total);
  std::vector<int32_t> ids((size_t)S * TOPK_MAX), cid((size_t)S * k);
  std::vector<float> lps((size_t)S * TOPK_MAX), clp((size_t)S * k);
  for (int depth = 0; depth < p->max_depth; depth++) {
    WB_HIP(hipMemcpy(st.data(), d_state.p, st.size() * 4, hipMemcpyDeviceToHost));
    const int n = st[ST_N];
    if (n == 0) break;
    WB_REQUIRE(n <= S, WB_ERR_STATE, "wb_beam_search_device: %d live rows", n);
    for (int i = 0; i < n; i++) { tok[i] = st[L.tok + i]; par[i] = st[L.parent + i]; win[i] = st[L.win + i]; }
    const int apply_mask = (P + depth) <= p->mask_until_len;
    WB_TRY(step(user, tok.data(), par.data(), win.data(), n, apply_mask, k, cid.data(), clp.data()));
    for (int i = 0; i < n; i++)
      for (int j = 0; j < k; j++) { ids[(size_t)i * TOPK_MAX + j] = cid[(size_t)i * k + j]; lps[(size_t)i * TOPK_MAX + j] = clp[(size_t)i * k + j]; }
    WB_HIP(hipMemcpy(d_topk.p, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
    WB_HIP(hipMemcpy(d_topk.as<int32_t>() + (size_t)S * TOPK_MAX, lps.data(), lps.size() * 4, hipMemcpyHostToDevice));
    launch_dec_beam_update(nullptr, a);
  }
  WB_HIP(hipMemcpy(ctl.data(), d_ctl.p, ctl.size() * 4, hipMemcpyDeviceToHost));
  WB_REQUIRE(ctl[BC_ERR] == 0, WB_ERR_STATE, "beam search: NaN log-probability (reference panics)");
  bool all_done = true;
  return beam_chain_extract(ctl, bl, prompt, P, out_tokens, row_stride, out_lens, &all_done);
}

extern "C" {

int wb_session_step(wb_session* s, const int32_t* new_tokens, const int32_t* parent, const int32_t* window, int n,
                    int apply_special_mask, int k, int32_t* top_ids, float* top_logprobs) {
  WB_REQUIRE(s && new_tokens && parent && window, WB_ERR_ARG, "wb_session_step: null argument");
  wb_model* m = s->m;
  const wb_dims& D = m->dims;
  const int V = D.n_vocab, S = s->S;
  wb::GpuTurn turn(s->device);
  WB_HIP(hipSetDevice(m->device));
  if (!s->decode_ready) WB_TRY(session_reserve(s, D.n_text_ctx));
  WB_REQUIRE(n >= 1 && n <= S, WB_ERR_ARG, "wb_session_step: n = %d outside [1, %d]", n, S);
  WB_REQUIRE(k >= 0 && k <= TOPK_MAX && (k == 0 || (top_ids && top_logprobs)), WB_ERR_ARG, "wb_session_step: bad k");
  WB_REQUIRE(!apply_special_mask || s->has_mask, WB_ERR_STATE, "wb_session_step: special mask not set");
  WB_REQUIRE(s->step < s->Lmax, WB_ERR_SHAPE, "Token sequence length %d must not exceed %d.", s->step + 1, s->Lmax);
  const StepLayout& L = s->lay;
  int* hs = s->state_host;
  memset(hs, 0, (size_t)L.total * 4);
  hs[ST_N] = n; hs[ST_STEP] = s->step;
  std::vector<int> len(n);
  for (int i = 0; i < n; i++) {
    WB_REQUIRE(new_tokens[i] >= 0 && new_tokens[i] < V, WB_ERR_ARG, "token id %d out of range [0,%d)", new_tokens[i], V);
    WB_REQUIRE(window[i] >= 0 && window[i] < s->W, WB_ERR_ARG, "beam %d: window %d out of range", i, window[i]);
    if (parent[i] < 0) {
      len[i] = 1;
    } else {
      WB_REQUIRE(parent[i] < s->prev_n, WB_ERR_ARG, "beam %d: parent %d is not a beam of the previous step", i, parent[i]);
      WB_REQUIRE(s->prev_win[parent[i]] == window[i], WB_ERR_ARG, "beam %d: parent belongs to another window", i);
      len[i] = s->prev_len[parent[i]] + 1;
    }
    WB_REQUIRE(len[i] <= s->Lmax, WB_ERR_SHAPE, "Token sequence length %d must not exceed %d.", len[i], s->Lmax);   // mod.rs:134-139
    hs[L.tok + i] = new_tokens[i]; hs[L.parent + i] = parent[i]; hs[L.len + i] = len[i]; hs[L.win + i] = window[i];
    int& nb = hs[L.win_nb + window[i]];
    WB_REQUIRE(nb < s->max_beams, WB_ERR_ARG, "window %d has more than max_beams = %d live beams", window[i], s->max_beams);
    hs[L.win_slots + window[i] * MAX_BEAMS + nb] = i;
    nb++;
  }
  // launch shape: bucketed so that the same captured graph serves every step of a decode.  9 - 16 live rows (the reference's
  // live setting: beam 5 over a 30 s chunk = 15 rows, transcribe.rs:232-233) stay on the fused sublayer kernels where those
  // exist (d <= 512): the attention / cross-attention blocks are per (head, row) anyway, the MLP block and the logits GEMV
  // run their 8-row tiles as two row groups -- 16 launches per step instead of batch mode's 54 (WHISPER_HIP_FUSE16=0: batch mode)
  static const bool fuse16_enabled = []() {
    const char* e = getenv("WHISPER_HIP_FUSE16"); const char* f = getenv("WHISPER_HIP_FUSE_SUB"); const char* x = getenv("WHISPER_HIP_FUSE_X");
    const char* c = getenv("WHISPER_HIP_FUSE_CO");
    return !(e && e[0] == '0') && !(f && f[0] == '0') && !(x && x[0] == '0') && !(c && c[0] == '1');
  }();
  const bool g16 = fuse16_enabled && n > 8 && n <= 16 && S >= 9 && dec_fused_supported(D.n_text_state) &&
                   D.n_text_state == 64 * D.n_text_head && s->maxC <= CROSS_FUSED_MAX_PASSES * CROSS_FUSED_MAX_C;
  const int n_launch = n <= 4 ? std::min(4, S) : n <= 8 ? std::min(8, S) : g16 ? std::min(16, S) : S;
  const bool fuse_ln = n_launch <= 8 || g16;      // LayerNorm in the GEMV prologue (redundant per block) vs its own launch
  const int max_nb = s->max_beams <= 1 ? 1 : s->max_beams <= 2 ? 2 : s->max_beams <= 4 ? 4 : 8;
  const int use_mask = apply_special_mask ? 1 : 0;
  hipStream_t st = s->st;
  const bool profiling = profile().on;
  static const bool graphs_enabled = []() { const char* e = getenv("WHISPER_HIP_GRAPH"); return !(e && e[0] == '0'); }();
  ScopedTimer tm_step(st, 3);
  WB_TRY(launch_step(s, n_launch, k, use_mask, fuse_ln, max_nb, graphs_enabled && !profiling, false, -1));
  tm_step.stop();
  WB_HIP(hipStreamSynchronize(st));   // results land in mapped host memory; state_host is reused by the next step
  WB_TRY(dec_split_check(s));
  s->last_had_logits = 0;
  if (k > 0) {
    for (int i = 0; i < n; i++)
      for (int j = 0; j < k; j++) {
        top_ids[i * k + j] = s->topk_id_host[i * TOPK_MAX + j];
        top_logprobs[i * k + j] = s->topk_lp_host[i * TOPK_MAX + j];
      }
    s->last_use_mask = use_mask;
    s->last_had_logits = 1;
  }
  tm_step.collect();
  if (tm_step.on) profile().ms[4] += 1;
  WB_HIP(hipGetLastError());
  s->prev_len = len;
  s->prev_win.assign(window, window + n);
  s->prev_n = n;
  s->step++;
  return WB_OK;
}

int wb_session_last_logprobs(wb_session* s, int slot, float* out) {
  WB_REQUIRE(s && out, WB_ERR_ARG, "wb_session_last_logprobs: null argument");
  WB_REQUIRE(s->last_had_logits && slot >= 0 && slot < s->prev_n, WB_ERR_STATE,
             "wb_session_last_logprobs: no logits for slot %d", slot);
  const int V = s->m->dims.n_vocab;
  wb::GpuTurn turn(s->device);
  WB_HIP(hipSetDevice(s->m->device));
  launch_dec_logprob_row(s->st, s->logits.as<float>() + (size_t)slot * V, 1, (int64_t)s->S * V, V,
                         s->mask.as<float>(), s->last_use_mask,
                         s->row_stats.as<float>() + 2 * slot, s->lp_tmp.as<float>());
  WB_HIP(hipMemcpyAsync(out, s->lp_tmp.p, (size_t)V * 4, hipMemcpyDeviceToHost, s->st));
  WB_HIP(hipStreamSynchronize(s->st));
  return WB_OK;
}

int wb_session_encoder_output(wb_session* s, int w, float* out, int32_t* C) {
  WB_REQUIRE(s && w >= 0 && w < s->W, WB_ERR_ARG, "wb_session_encoder_output: bad argument");
  const int d = s->m->dims.n_audio_state;
  if (C) *C = s->C[w];
  if (out) {
    wb::GpuTurn turn(s->device);
    WB_HIP(hipSetDevice(s->m->device));
    WB_HIP(hipMemcpyAsync(out, s->enc_out.as<float>() + (size_t)s->row0[w] * d, (size_t)s->C[w] * d * 4,
                          hipMemcpyDeviceToHost, s->st));
    WB_HIP(hipStreamSynchronize(s->st));
  }
  return WB_OK;
}

int wb_profile_enable(int on) {
  profile().on = on != 0;
  return WB_OK;
}
int wb_profile_kernels(wb_kernel_stat* out, int cap, int reset) {
  WB_REQUIRE(out || cap == 0, WB_ERR_ARG, "wb_profile_kernels: null argument");
  prof_collect();
  int n = 0;
  for (int c = 0; c < KC_COUNT; c++) {
    if (g_kstats[c].calls == 0) continue;
    if (n < cap) {
      snprintf(out[n].name, sizeof(out[n].name), "%s", g_kernel_names[c]);
      out[n].calls = g_kstats[c].calls; out[n].total_ms = g_kstats[c].ms; out[n].algo_bytes = g_kstats[c].bytes;
    }
    n++;
  }
  if (reset)
    for (int c = 0; c < KC_COUNT; c++) g_kstats[c] = KernelStat();
  return n;
}
int wb_profile_read(double* out8, int reset) {
  WB_REQUIRE(out8, WB_ERR_ARG, "wb_profile_read: null argument");
  for (int i = 0; i < 8; i++) out8[i] = profile().ms[i];
  if (reset)
    for (int i = 0; i < 8; i++) profile().ms[i] = 0;
  return WB_OK;
}

}  // extern "C"
