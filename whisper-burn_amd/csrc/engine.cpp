// Encoder / decoder forward passes: the reference's module graph (src/model/mod.rs) expressed as a
// short sequence of fused gfx950 launches over packed, ragged window batches.
#include "engine.h"

#include <sys/syscall.h>
#include <unistd.h>

#include <functional>
#include <map>
#include <mutex>
#include <string>

namespace wb {

GpuTurn::GpuTurn(int device) {
  static std::recursive_mutex mu[64];          // (per device: the fault is a co-execution fault of one GPU's SIMDs)
  static const bool enabled = []() { const char* e = getenv("WHISPER_HIP_GPU_TURN"); return !(e && e[0] == '0'); }();
  if (enabled) lk = std::unique_lock<std::recursive_mutex>(mu[device < 0 ? 0 : device & 63]);
}

int get_mel_tables(int device, double sample_rate, const MelTables** out_dev) {
  static std::mutex mu;
  static std::map<std::pair<int, double>, MelTables*> cache;
  std::lock_guard<std::mutex> lk(mu);
  auto key = std::make_pair(device, sample_rate);
  auto it = cache.find(key);
  if (it != cache.end()) { *out_dev = it->second; return WB_OK; }
  auto host = std::make_unique<MelTables>();
  WB_REQUIRE(mel_tables_build(sample_rate, host.get()) == 0, WB_ERR_SHAPE,
             "mel filterbank for sample_rate %g has a row wider than %d taps", sample_rate, MEL_MAX_TAPS);
  MelTables* dev = nullptr;
  WB_HIP(hipSetDevice(device));
  WB_HIP(hipMalloc(&dev, sizeof(MelTables)));
  {  // (once per (device, rate); on a stream of its own: a legacy-stream copy fails while another thread captures a step graph)
    hipStream_t ts = nullptr;
    WB_HIP(hipStreamCreateWithFlags(&ts, hipStreamNonBlocking));
    const hipError_t e1 = hipMemcpyAsync(dev, host.get(), sizeof(MelTables), hipMemcpyHostToDevice, ts);
    const hipError_t e2 = hipStreamSynchronize(ts);
    (void)hipStreamDestroy(ts);
    WB_HIP(e1);
    WB_HIP(e2);
  }
  cache[key] = dev;
  *out_dev = dev;
  return WB_OK;
}

// Developer tool (WHISPER_HIP_ENC_TRACE=<dir>): stream-ordered device copies of every stage of an encoder pass, written to
// <dir>/enc_trace_<tid>.bin at the end of the pass (one extra synchronisation there, none inside) -- to find the first stage
// whose result differs between two runs (tools/probe_threads_enc.py).
namespace {
struct EncTrace { std::vector<std::string> name; std::vector<DevMem> buf; std::vector<size_t> bytes; size_t n = 0; };
thread_local EncTrace tl_trace;
const char* enc_trace_dir() { static const char* d = getenv("WHISPER_HIP_ENC_TRACE"); return d; }
void trace_stage(hipStream_t st, const std::string& name, const void* p, size_t bytes) {
  if (!enc_trace_dir()) return;
  EncTrace& t = tl_trace;
  if (t.n == t.buf.size()) { t.buf.emplace_back(); t.name.emplace_back(); t.bytes.push_back(0); }
  if (t.buf[t.n].ensure(bytes) != WB_OK) return;
  (void)hipMemcpyAsync(t.buf[t.n].p, p, bytes, hipMemcpyDeviceToDevice, st);
  t.name[t.n] = name; t.bytes[t.n] = bytes; t.n++;
}
void trace_flush(hipStream_t st) {
  if (!enc_trace_dir()) return;
  EncTrace& t = tl_trace;
  (void)hipStreamSynchronize(st);
  char path[512];
  snprintf(path, sizeof(path), "%s/enc_trace_%ld.bin", enc_trace_dir(), (long)syscall(SYS_gettid));
  FILE* f = fopen(path, "wb");
  std::vector<char> host;
  for (size_t i = 0; f && i < t.n; i++) {
    host.resize(t.bytes[i]);
    (void)hipMemcpy(host.data(), t.buf[i].p, t.bytes[i], hipMemcpyDeviceToHost);
    char nm[32] = {0};
    snprintf(nm, sizeof(nm), "%s", t.name[i].c_str());
    const int64_t nb = (int64_t)t.bytes[i];
    fwrite(nm, 1, 32, f); fwrite(&nb, 8, 1, f); fwrite(host.data(), 1, host.size(), f);
  }
  if (f) fclose(f);
  t.n = 0;
}
}  // namespace
void enc_trace_stage(hipStream_t st, const char* name, const void* p, size_t bytes) { trace_stage(st, name, p, bytes); }

static int upload(hipStream_t st, DevMem& dst, const void* src, size_t bytes) {
  WB_TRY(dst.ensure(bytes));
  WB_HIP(hipMemcpyAsync(dst.p, src, bytes, hipMemcpyHostToDevice, st));
  return WB_OK;
}

// Dispatch: the split-precision fp16 MFMA kernel when the weight carries split copies (sh, sl: [N][ldwt] fp16) and the shape
// fits it, else the exact-f32 MFMA kernel (the conv1 gather, split-K launches, decoder-side weights).
// The range-flag word of the guarded pass THIS thread is enqueueing (device-visible address of a mapped host word; null:
// no guarded pass -> the exact-f32 kernel).  A pass is enqueued synchronously by one host thread, so a thread-local scope
// reaches every gemm_dispatch of the pass without threading a pointer through the whole call tree.
static thread_local int* tl_split_flag = nullptr;
namespace {
struct SplitFlagScope {
  int* prev;
  explicit SplitFlagScope(int* f) : prev(tl_split_flag) { tl_split_flag = f; }
  ~SplitFlagScope() { tl_split_flag = prev; }
};
}  // namespace

int gemm_dispatch(const wb_model* m, hipStream_t st, const GemmArgs& a, int ldwt, const uint16_t* sh, const uint16_t* sl) {
  // exact-f32 models: encoder-side weights that carry a split copy go through the three-product fp16 kernel
  // (a.Ah: the activations already ARE fp16 pieces -- the producer decided at the top of the pass; a model that left the split
  // kernel in the meantime (another session's guard tripped) still finishes this pass on it, judged by the pass's own flag)
  const bool pieces = a.Ah != nullptr || a.Ch != nullptr;
  if ((m->split_active() || pieces) && tl_split_flag && sh && sl && a.conv1_tstride == 0 && a.K % 32 == 0 && ldwt % 8 == 0 &&
      a.ksplit <= 1 && (a.a_desc == nullptr || a.a_mask_align % 8 == 0)) {
    GemmArgs g = a;
    g.range_flag = tl_split_flag;
    WB_REQUIRE(launch_gemm_f16x3(st, g, sh, sl, ldwt) == 0, WB_ERR_SHAPE, "split gemm: unsupported shape M=%d N=%d K=%d", a.M,
               a.N, a.K);
    return WB_OK;
  }
  WB_REQUIRE(!pieces, WB_ERR_STATE, "gemm: activation pieces handed to a GEMM that cannot take the split-precision kernel");
  WB_REQUIRE(launch_gemm_f32(st, a) == 0, WB_ERR_SHAPE, "gemm: unsupported shape M=%d N=%d K=%d ldb=%d", a.M,
             a.N, a.K, a.ldb);
  return WB_OK;
}

// Run `body` (a pass that may use the split-precision kernel); if that kernel raised its range flag -- an activation left
// fp16's range, so some outputs are inf / NaN -- switch the model to the exact-f32 kernel for good and run the pass again.
// The flag word belongs to the PASS (`flag_host` / `flag_dev`: a session's own mapped word, or the model's for the
// stateless entry points, which api.cpp serialises): sessions of one model on different streams and threads share
// nothing here -- no lock, and a pass can neither see nor clear a flag another pass raised.  split_off is the only shared
// state (0 -> 1, once, atomic): a pass that is being enqueued while another one trips the guard runs the rest of its GEMMs
// on the exact-f32 kernel, which is valid in any mix, and is judged by its own flag.
// `deferred` (null: check now, at the cost of one stream synchronisation): the pass is enqueued and *deferred = true;
// the caller reads the flag at its next natural synchronisation with split_guard_resolve() and repeats the work itself.
int split_guarded(wb_model* m, hipStream_t st, int* flag_host, int* flag_dev, bool* deferred,
                  const std::function<int()>& body) {
  if (deferred) *deferred = false;
  if (!m->split_active() || !flag_host || !flag_dev) { SplitFlagScope off(nullptr); return body(); }
  int rc;
  { SplitFlagScope on(flag_dev); rc = body(); }
  if (rc != WB_OK) {
    // kernels of the failed pass may still raise the flag: drain them and leave the word clear for the next pass
    (void)hipStreamSynchronize(st);
    __atomic_store_n(flag_host, 0, __ATOMIC_RELEASE);
    return rc;
  }
  if (deferred) { *deferred = true; return WB_OK; }
  WB_HIP(hipStreamSynchronize(st));
  if (!split_guard_resolve(m, flag_host)) return WB_OK;
  SplitFlagScope off(nullptr);
  return body();
}

// After the stream the guarded pass ran on has been synchronised: did the pass raise its flag?  If so the word is cleared,
// the model leaves the split-precision kernel for good and the caller must repeat the pass (and whatever consumed its output).
bool split_guard_resolve(wb_model* m, int* flag_host) {
  if (!flag_host || __atomic_load_n(flag_host, __ATOMIC_ACQUIRE) == 0) return false;
  __atomic_store_n(flag_host, 0, __ATOMIC_RELEASE);
  __atomic_store_n(&m->split_off, 1, __ATOMIC_RELEASE);
  return true;
}

// GEMM against a model weight: `w` supplies the split copies (null: f32 kernel only).
static int gemm(const wb_model* m, hipStream_t st, const GemmArgs& a, const LinearW* w) {
  return gemm_dispatch(m, st, a, w ? w->k : 0, w ? w->sh : nullptr, w ? w->sl : nullptr);
}

// y = x + (h W + b) for a long contraction (the MLP's second matrix, K = 4 d): K in blocks of GEMM_KBLOCK rows, each block's
// product added to the running result in place.  The exact-f32 MFMA GEMM accumulates K sequentially in ONE f32 chain per
// output; over K = 3072 / 5120 that chain is 2 - 2.8 x less accurate than the blocked sums of a CPU BLAS (measured against
// f64), and with 24 - 64 such products per forward it set the distance of the whole path from the
// exact result.  Blocks of 1024 cost one extra read + write of the [M][d] result per block (~1.5 % of the encoder).
constexpr int GEMM_KBLOCK = 1024;
// Ah / Al non-null: A as fp16 pieces (GemmArgs::Ah) -- only when every block takes the split-precision kernel (the caller
// checks w.sh and the shape)
static int gemm_residual_kblocked(const wb_model* m, hipStream_t st, const float* A, int M, const LinearW& w, float* x,
                                  const uint16_t* Ah = nullptr, const uint16_t* Al = nullptr) {
  const bool blocked = w.k >= 2 * GEMM_KBLOCK && w.k % GEMM_KBLOCK == 0;
  const int kb = blocked ? GEMM_KBLOCK : w.k;
  for (int k0 = 0; k0 < w.k; k0 += kb) {
    // In-place accumulation (C == residual == x): sound only because every GEMM kernel reads residual[row][col] and writes
    // C[row][col] from the SAME thread, once, with ksplit <= 1 (GemmArgs::residual: "may alias C") -- a split-K or
    // multi-pass tile configuration would break it, hence the explicit ksplit below and the check in gemm_dispatch's callee.
    GemmArgs g;
    g.A = A + k0; g.lda = w.k; g.B = w.w + (size_t)k0 * w.n;
    if (Ah) { g.Ah = Ah + k0; g.Al = Al + k0; } g.ldb = w.n; g.C = x; g.ldc = w.n;
    g.bias = k0 == 0 ? w.b : nullptr;              // first block: + bias + the residual stream; later blocks: + the running sum
    g.residual = x; g.ldr = w.n;
    g.M = M; g.N = w.n; g.K = kb; g.ksplit = 1;
    WB_REQUIRE(g.residual == g.C && g.ksplit <= 1, WB_ERR_STATE, "k-blocked residual GEMM: aliasing contract");
    WB_TRY(gemm_dispatch(m, st, g, w.k, w.sh ? w.sh + k0 : nullptr, w.sl ? w.sl + k0 : nullptr));
  }
  return WB_OK;
}

static GemmArgs linear_args(const float* A, int M, const LinearW& w, float* C) {
  GemmArgs g;
  g.A = A; g.lda = w.k; g.B = w.w; g.ldb = w.n; g.C = C; g.ldc = w.n; g.bias = w.b;
  g.M = M; g.N = w.n; g.K = w.k;
  return g;
}

int run_encoder(wb_model* m, hipStream_t st, Workspace& ws, const MelBatch& mb, float* out_dev, EncoderOut* eo) {
  // (stateless entry: the model's own flag word; api.cpp serialises these calls)
  return split_guarded(m, st, m->split_flag_host, m->split_flag_dev, nullptr,
                       [&]() { return run_encoder_unguarded(m, st, ws, mb, out_dev, eo); });
}

int run_encoder_unguarded(wb_model* m, hipStream_t st, Workspace& ws, const MelBatch& mb, float* out_dev, EncoderOut* eo) {
  const wb_dims& D = m->dims;
  const int d = D.n_audio_state, H = D.n_audio_head, nw = (int)mb.T.size();
  WB_REQUIRE(nw > 0, WB_ERR_ARG, "encoder: empty batch");
  eo->C.resize(nw); eo->row0.resize(nw);
  int rows1 = 0, rows2 = 0, maxC = 0;
  std::vector<int> x1row0(nw);
  for (int w = 0; w < nw; w++) {
    // mod.rs:236-241 (the reference bounds MEL FRAMES by n_audio_ctx)
    WB_REQUIRE(mb.T[w] >= 1 && mb.T[w] <= m->max_mel_frames(), WB_ERR_SHAPE, "Audio length %d cannot exceed %d.",
               mb.T[w], m->max_mel_frames());
    x1row0[w] = rows1; rows1 += mb.T[w];
    eo->C[w] = (mb.T[w] - 1) / 2 + 1;
    eo->row0[w] = rows2; rows2 += eo->C[w];
    maxC = std::max(maxC, eo->C[w]);
  }
  eo->rows = rows2;
  // ---- per-row descriptors of the two implicit-GEMM convolutions (cached per geometry) ----
  const bool same_geom = ws.enc_T == mb.T && ws.enc_win_stride == mb.win_stride && ws.enc_row_stride == mb.row_stride &&
                         ws.enc_d == d && ws.desc1.p && ws.desc2.p && ws.auxidx.p && ws.segs.p;
  if (!same_geom) {
  ws.enc_T.clear(); ws.enc_d = -1;    // the key names the device contents: void it BEFORE they change, set it after the sync
  std::vector<RowDesc> d1(rows1), d2(rows2);
  std::vector<int32_t> aidx(rows2);
  std::vector<AttnSeg> segs(nw);
  for (int w = 0; w < nw; w++) {
    const int T = mb.T[w], C = eo->C[w];
    for (int t = 0; t < T; t++) {
      RowDesc& r = d1[x1row0[w] + t];
      r.off = (int32_t)(w * mb.win_stride + t); r.klo = (t == 0); r.khi = (t == T - 1);
    }
    for (int c = 0; c < C; c++) {
      RowDesc& r = d2[eo->row0[w] + c];
      r.off = (int32_t)(((int64_t)x1row0[w] + 2 * c - 1) * d);   // rows 2c-1, 2c, 2c+1 of x1 are contiguous
      r.klo = (int16_t)(c == 0 ? d : 0);                         // zero padding left of the window
      r.khi = (int16_t)(2 * c + 1 >= T ? 2 * d : 3 * d);         // ... and right of it
      aidx[eo->row0[w] + c] = c;                                 // positional_embedding[0..C] (mod.rs:247-252)
    }
    segs[w] = AttnSeg{eo->row0[w], C, eo->row0[w], C};
  }
  WB_REQUIRE((int64_t)nw * mb.win_stride < (int64_t)1 << 31 && (int64_t)rows1 * d < (int64_t)1 << 31, WB_ERR_SHAPE,
             "encoder batch too large for 32-bit row offsets");
  WB_TRY(upload(st, ws.desc1, d1.data(), d1.size() * sizeof(RowDesc)));
  WB_TRY(upload(st, ws.desc2, d2.data(), d2.size() * sizeof(RowDesc)));
  WB_TRY(upload(st, ws.auxidx, aidx.data(), aidx.size() * sizeof(int32_t)));
  WB_TRY(upload(st, ws.segs, segs.data(), segs.size() * sizeof(AttnSeg)));
  WB_HIP(hipStreamSynchronize(st));   // host vectors die at scope exit
  ws.enc_T = mb.T; ws.enc_win_stride = mb.win_stride; ws.enc_row_stride = mb.row_stride; ws.enc_d = d;
  }
  WB_TRY(ws.x1.ensure((size_t)rows1 * d * 4));
  WB_TRY(ws.x.ensure((size_t)rows2 * d * 4));
  WB_TRY(ws.h.ensure((size_t)rows2 * d * 4));
  WB_TRY(ws.qkv.ensure((size_t)rows2 * 3 * d * 4));
  WB_TRY(ws.att.ensure((size_t)rows2 * d * 4));
  WB_TRY(ws.hm.ensure((size_t)rows2 * 4 * d * 4));
  float *x1 = ws.x1.as<float>(), *x = ws.x.as<float>(), *h = ws.h.as<float>(), *qkv = ws.qkv.as<float>(),
        *att = ws.att.as<float>(), *hm = ws.hm.as<float>();

  // conv1 + GELU (mod.rs:243): implicit GEMM over [T, 240] x [240, d], output position-major [T, d]
  {
    GemmArgs g;
    g.A = mb.mel; g.a_desc = ws.desc1.as<RowDesc>(); g.conv1_tstride = mb.row_stride;
    g.B = m->conv1.w; g.ldb = d; g.C = x1; g.ldc = d; g.bias = m->conv1.b;
    g.M = rows1; g.N = d; g.K = 240; g.act = ACT_GELU;
    trace_stage(st, "mel", mb.mel, (size_t)nw * mb.win_stride * 4);
    WB_TRY(gemm(m, st, g, nullptr));
    trace_stage(st, "conv1", x1, (size_t)rows1 * d * 4);
  }
  // conv2 (stride 2) + GELU + transpose + positional add (mod.rs:244-252): rows 2c-1..2c+1 of x1 form one A row
  {
    GemmArgs g;
    g.A = x1; g.a_desc = ws.desc2.as<RowDesc>(); g.a_mask_align = d;     // (klo in {0, d}, khi in {2 d, 3 d})
    g.B = m->conv2.w; g.ldb = d; g.C = x; g.ldc = d; g.bias = m->conv2.b;
    g.M = rows2; g.N = d; g.K = 3 * d; g.act = ACT_GELU;
    g.aux = m->enc_pos; g.aux_idx = ws.auxidx.as<int32_t>(); g.ld_aux = d;
    WB_TRY(gemm(m, st, g, &m->conv2));
    trace_stage(st, "conv2", x, (size_t)rows2 * d * 4);
  }
  // Activation pieces (round 6): inside a guarded split-precision pass the PRODUCERS of the layer GEMMs' A operands --
  // LayerNorm, the attention kernel, the GELU epilogue of lin1 -- write them as fp16 hi / lo planes once (the same 4 bytes per
  // element, in the same buffers), and the GEMM's A path is two 16-byte loads straight to LDS.  WHISPER_HIP_ENCODER_PIECES=0
  // keeps f32 activations (every column block of every consumer then splits its A tile itself, as in round 5).
  static const bool pieces_enabled = []() { const char* e = getenv("WHISPER_HIP_ENCODER_PIECES"); return !(e && e[0] == '0'); }();
  const bool split_pass = m->split_active() && tl_split_flag != nullptr && d % 32 == 0;
  uint16_t* h_hi = reinterpret_cast<uint16_t*>(h); uint16_t* h_lo = h_hi + (size_t)rows2 * d;
  uint16_t* att_hi = reinterpret_cast<uint16_t*>(att); uint16_t* att_lo = att_hi + (size_t)rows2 * d;
  uint16_t* hm_hi = reinterpret_cast<uint16_t*>(hm); uint16_t* hm_lo = hm_hi + (size_t)rows2 * 4 * d;
  for (int i = 0; i < D.n_audio_layer; i++) {   // ResidualEncoderAttentionBlock::forward, mod.rs:299-303
    const EncBlockW& b = m->enc[i];
    const bool pcs = pieces_enabled && split_pass && b.qkv.sh && b.out.sh && b.mlp1.sh && b.mlp2.sh;
    GemmArgs g = linear_args(h, rows2, b.qkv, qkv);
    if (pcs) {
      launch_layernorm_pieces(st, x, h_hi, h_lo, rows2, d, b.ln1.g, b.ln1.b, b.ln1.eps, m->ln_eps_inside_sqrt);
      g.Ah = h_hi; g.Al = h_lo;
    } else {
      launch_layernorm(st, x, h, rows2, d, b.ln1.g, b.ln1.b, b.ln1.eps, m->ln_eps_inside_sqrt);
    }
    g.col_scale = m->qk_scale; g.col_scale_period = 3 * d; g.col_scale_width = 2 * d;   // q*s, k*s (mod.rs:506-514)
    const std::string L = "L" + std::to_string(i) + ".";
    trace_stage(st, L + "ln1", h, (size_t)rows2 * d * 4);
    WB_TRY(gemm(m, st, g, &b.qkv));
    trace_stage(st, L + "qkv", qkv, (size_t)rows2 * 3 * d * 4);
    // (split precision only inside a guarded pass whose out-projection runs on the split GEMM: that GEMM's range guard is
    // what reports an attention operand outside fp16's range -- NaN in, flag raised, the pass repeated in exact f32)
    const bool att_pcs = launch_attention(st, qkv, 3 * d, qkv + d, qkv + 2 * d, 3 * d, att, d, ws.segs.as<AttnSeg>(), nw, maxC,
                                          H, 1.0f, 0, split_pass && b.out.sh != nullptr, pcs ? att_hi : nullptr,
                                          pcs ? att_lo : nullptr);
    trace_stage(st, L + "att", att, (size_t)rows2 * d * 4);
    g = linear_args(att, rows2, b.out, x);
    if (att_pcs) { g.Ah = att_hi; g.Al = att_lo; }
    g.residual = x; g.ldr = d;
    WB_TRY(gemm(m, st, g, &b.out));
    trace_stage(st, L + "out", x, (size_t)rows2 * d * 4);
    g = linear_args(h, rows2, b.mlp1, hm);
    if (pcs) {
      launch_layernorm_pieces(st, x, h_hi, h_lo, rows2, d, b.ln2.g, b.ln2.b, b.ln2.eps, m->ln_eps_inside_sqrt);
      g.Ah = h_hi; g.Al = h_lo;
      g.Ch = hm_hi; g.Cl = hm_lo;                 // GELU(lin1) as pieces: lin2's A operand
    } else {
      launch_layernorm(st, x, h, rows2, d, b.ln2.g, b.ln2.b, b.ln2.eps, m->ln_eps_inside_sqrt);
    }
    g.act = ACT_GELU;
    trace_stage(st, L + "ln2", h, (size_t)rows2 * d * 4);
    WB_TRY(gemm(m, st, g, &b.mlp1));
    trace_stage(st, L + "mlp1", hm, (size_t)rows2 * 4 * d * 4);
    WB_TRY(gemm_residual_kblocked(m, st, hm, rows2, b.mlp2, x, pcs ? hm_hi : nullptr, pcs ? hm_lo : nullptr));
    trace_stage(st, L + "mlp2", x, (size_t)rows2 * d * 4);
  }
  launch_layernorm(st, x, out_dev, rows2, d, m->ln_post.g, m->ln_post.b, m->ln_post.eps, m->ln_eps_inside_sqrt);
  trace_stage(st, "ln_post", out_dev, (size_t)rows2 * d * 4);
  trace_flush(st);
  WB_HIP(hipGetLastError());
  return WB_OK;
}

static int run_decoder_stateless_body(wb_model* m, hipStream_t st, Workspace& ws, const int32_t* tokens_dev, int n, int L,
                                      const float* enc_dev, int C, float* logits_dev);
int run_decoder_stateless(wb_model* m, hipStream_t st, Workspace& ws, const int32_t* tokens_dev, int n, int L,
                          const float* enc_dev, int C, float* logits_dev) {
  // (guarded: the cross-K/V projection of every layer, ckv_all, runs on the split-precision kernel here too)
  return split_guarded(m, st, m->split_flag_host, m->split_flag_dev, nullptr,
                       [&]() { return run_decoder_stateless_body(m, st, ws, tokens_dev, n, L, enc_dev, C, logits_dev); });
}

static int run_decoder_stateless_body(wb_model* m, hipStream_t st, Workspace& ws, const int32_t* tokens_dev, int n, int L,
                                      const float* enc_dev, int C, float* logits_dev) {
  const wb_dims& D = m->dims;
  const int d = D.n_text_state, H = D.n_text_head, NL = D.n_text_layer, V = D.n_vocab;
  const int rows = n * L, krows = n * C, ldkv = NL * 2 * d;
  std::vector<AttnSeg> segs(2 * n);
  for (int i = 0; i < n; i++) {
    segs[i] = AttnSeg{i * L, L, i * L, L};          // masked self-attention
    segs[n + i] = AttnSeg{i * L, L, i * C, C};      // cross-attention
  }
  WB_TRY(upload(st, ws.segs, segs.data(), segs.size() * sizeof(AttnSeg)));
  WB_HIP(hipStreamSynchronize(st));
  ws.enc_T.clear();                   // (ws.segs no longer holds the encoder's segments)
  WB_TRY(ws.x.ensure((size_t)rows * d * 4));
  WB_TRY(ws.h.ensure((size_t)rows * d * 4));
  WB_TRY(ws.qkv.ensure((size_t)rows * 3 * d * 4));
  WB_TRY(ws.att.ensure((size_t)rows * d * 4));
  WB_TRY(ws.hm.ensure((size_t)rows * 4 * d * 4));
  WB_TRY(ws.x1.ensure((size_t)krows * ldkv * 4));   // cross K|V of every layer
  float *x = ws.x.as<float>(), *h = ws.h.as<float>(), *qkv = ws.qkv.as<float>(), *att = ws.att.as<float>(),
        *hm = ws.hm.as<float>(), *ckv = ws.x1.as<float>();
  const AttnSeg* sg = ws.segs.as<AttnSeg>();

  launch_embed(st, tokens_dev, rows, L, d, m->tok_emb, m->dec_pos, x);   // mod.rs:141-146
  // cross-attention K/V (mod.rs:484-485) for all layers in one GEMM; K columns pre-scaled (mod.rs:510-514)
  GemmArgs g = linear_args(enc_dev, krows, m->ckv_all, ckv);
  g.col_scale = m->qk_scale; g.col_scale_period = 2 * d; g.col_scale_width = d;
  WB_TRY(gemm(m, st, g, &m->ckv_all));
  for (int i = 0; i < NL; i++) {   // ResidualDecoderAttentionBlock::forward, mod.rs:345-350
    const DecBlockW& b = m->dec[i];
    launch_layernorm(st, x, h, rows, d, b.ln1.g, b.ln1.b, b.ln1.eps, m->ln_eps_inside_sqrt);
    g = linear_args(h, rows, b.qkv, qkv);
    g.col_scale = m->qk_scale; g.col_scale_period = 3 * d; g.col_scale_width = 2 * d;
    WB_TRY(gemm(m, st, g, &b.qkv));
    launch_attention_f32(st, qkv, 3 * d, qkv + d, qkv + 2 * d, 3 * d, att, d, sg, n, L, H, 1.0f, 1);
    g = linear_args(att, rows, b.out, x);
    g.residual = x; g.ldr = d;
    WB_TRY(gemm(m, st, g, &b.out));
    launch_layernorm(st, x, h, rows, d, b.ln2.g, b.ln2.b, b.ln2.eps, m->ln_eps_inside_sqrt);
    g = linear_args(h, rows, b.cq, qkv);
    g.col_scale = m->qk_scale; g.col_scale_period = d; g.col_scale_width = d;
    WB_TRY(gemm(m, st, g, &b.cq));
    launch_attention_f32(st, qkv, d, ckv + (size_t)i * 2 * d, ckv + (size_t)i * 2 * d + d, ldkv, att, d, sg + n, n,
                         L, H, 1.0f, 0);
    g = linear_args(att, rows, b.cout, x);
    g.residual = x; g.ldr = d;
    WB_TRY(gemm(m, st, g, &b.cout));
    launch_layernorm(st, x, h, rows, d, b.ln3.g, b.ln3.b, b.ln3.eps, m->ln_eps_inside_sqrt);
    g = linear_args(h, rows, b.mlp1, hm);
    g.act = ACT_GELU;
    WB_TRY(gemm(m, st, g, &b.mlp1));
    WB_TRY(gemm_residual_kblocked(m, st, hm, rows, b.mlp2, x));
  }
  launch_layernorm(st, x, h, rows, d, m->ln_dec.g, m->ln_dec.b, m->ln_dec.eps, m->ln_eps_inside_sqrt);
  // logits = x . token_embedding^T (mod.rs:156), streamed from the [d][Vp] transposed copy
  GemmArgs lg;
  lg.A = h; lg.lda = d; lg.B = m->tok_emb_t; lg.ldb = m->vocab_ld; lg.C = logits_dev; lg.ldc = V;
  lg.M = rows; lg.N = V; lg.K = d;
  WB_TRY(gemm_dispatch(m, st, lg, 0));
  WB_HIP(hipGetLastError());
  return WB_OK;
}

}  // namespace wb
