// Internal declarations shared by the host side of libwhisper_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <memory>
#include <string>
#include <unordered_map>
#include <mutex>
#include <vector>

#include "../../include/whisper_hip.h"

namespace wb {

// One GPU "turn" per device and process.  Every entry point that enqueues kernels holds its device's turn from its first launch
// to its last synchronisation, so kernels of different sessions / streams of this process never run on one GPU at the same time.
// Why (round 6, profiles/r06_y_*): on gfx950 a wave's packed-FP32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32)
// return wrong results in lanes 48-63 while a wave of ANOTHER kernel executes f16 MFMAs (v_mfma_f32_32x32x16_f16,
// v_mfma_f32_16x16x32_f16) on the same SIMD -- tools/pk_mfma_probe.cpp reproduces it without this library.  Two threads
// transcribing with one model hit it as a log-mel whose DFT butterflies were wrong in a few frames (the other thread's
// split-precision encoder GEMM next to the mel kernel) and, from there, different tokens.  Within a call every kernel is on one
// stream, in order; the turn keeps calls from overlapping.  Recursive: wb_waveform_to_tokens -> wb_session_decode.
// WHISPER_HIP_GPU_TURN=0 (developer switch: the probes that demonstrate the fault) makes it a no-op.
struct GpuTurn {
  std::unique_lock<std::recursive_mutex> lk;
  explicit GpuTurn(int device);
};

// ---- errors -------------------------------------------------------------------
void set_error(const char* fmt, ...);
const char* get_error();

#define WB_HIP(expr)                                                                   \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      wb::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,   \
                    __LINE__);                                                         \
      return (_e == hipErrorOutOfMemory) ? WB_ERR_OOM : WB_ERR_HIP;                    \
    }                                                                                  \
  } while (0)

#define WB_TRY(expr)              \
  do {                            \
    int _s = (expr);              \
    if (_s != WB_OK) return _s;   \
  } while (0)

#define WB_REQUIRE(cond, status, ...) \
  do {                                \
    if (!(cond)) {                    \
      wb::set_error(__VA_ARGS__);     \
      return (status);                \
    }                                 \
  } while (0)

// ---- device memory ------------------------------------------------------------
// Owning device allocation; freed with the owner (model / session / scratch arena).
struct DevMem {
  void* p = nullptr;
  size_t bytes = 0;
  DevMem() = default;
  DevMem(const DevMem&) = delete;
  DevMem& operator=(const DevMem&) = delete;
  DevMem(DevMem&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
  DevMem& operator=(DevMem&& o) noexcept {
    if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
    return *this;
  }
  ~DevMem() { release(); }
  int alloc(size_t n);           // returns wb_status
  int ensure(size_t n);          // grow-only
  // grow-only, and a fresh allocation is zero-filled: for buffers a kernel may READ before it has written them under
  // a zero weight (cached K/V rows past the current position enter a dot product with probability 0) -- whatever bit
  // patterns hipMalloc hands back, including NaN / Inf, must not reach the arithmetic
  int ensure_zeroed(size_t n, hipStream_t st = nullptr);   // (st: the owner's stream -- never the legacy stream when sessions run concurrently)
  void release();
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// ---- host tensors (loader input) ------------------------------------------------
struct HostTensor {
  std::vector<int64_t> shape;
  const float* data = nullptr;          // borrowed or -> owned
  std::vector<float> owned;
  int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};
using TensorMap = std::unordered_map<std::string, HostTensor>;

int read_dump_dir(const char* dir, TensorMap* out);   // model_load.cpp
int read_burn_record(const char* mpk_gz_path, const char* cfg_path, TensorMap* out);   // record_load.cpp
}  // namespace wb
struct wb_model;
namespace wb {
int build_model(TensorMap& tm, int device, int compute_dtype, wb_model** out);

// ---- model ----------------------------------------------------------------------
struct LayerNormW { float* g = nullptr; float* b = nullptr; float eps = 1e-5f; };
struct LinearW {
  float* w = nullptr; float* b = nullptr; int k = 0, n = 0;   // w: [k][n] row-major f32
  uint16_t* sh = nullptr; uint16_t* sl = nullptr;   // split-precision path: fp16 hi / lo * 2^11, [n][k] (encoder-side weights)
  // decoder-side weights, batch-mode decode (decode_batch.hip, dec_skinny_f16x3_kernel): the same two pieces in MFMA tiles,
  // [n / 16][k / 32][(k % 32) / 8][n % 16][k % 8] -- a wave's B operand of one (16-column, 32-deep) tile is 1 KB contiguous
  uint16_t* th = nullptr; uint16_t* tl = nullptr;
};

struct EncBlockW {
  LayerNormW ln1, ln2;
  LinearW qkv;   // [d][3d], bias [3d] (key part zero: mod.rs:402-404)
  LinearW out;   // [d][d]
  LinearW mlp1;  // [d][4d]
  LinearW mlp2;  // [4d][d]
};
struct DecBlockW {
  LayerNormW ln1, ln2, ln3;
  LinearW qkv, out;        // masked self-attention
  LinearW cq, cout;        // cross-attention query / out
  LinearW mlp1, mlp2;
};

// Grow-only device workspace shared by the forward passes of one owner (model scratch or session).
struct Workspace {
  DevMem x1, x, h, qkv, att, hm, desc1, desc2, auxidx, segs, misc;
  // geometry the encoder descriptors on the device were built for: a transcription loop over equally shaped batches
  // (the common case) re-uses them instead of paying four uploads and a stream synchronisation per batch
  std::vector<int> enc_T; int64_t enc_win_stride = -1; int enc_row_stride = -1; int enc_d = -1;
};

}  // namespace wb

// Encoder-side Linear layers of exact-f32 models run on the split-precision kernel (gemm_f16x3.hip) unless
// WHISPER_HIP_ENCODER_SPLIT=0: as close to the exact result as the f32 MFMA kernel or closer (its K chain is 16 x shorter:
// profiles/r04_a_diag_*), at about twice the rate.
constexpr bool WB_ENCODER_SPLIT_DEFAULT = true;

struct wb_model {
  // process-unique, never reused (build_model): the session pool and the captured decode graphs are keyed by it, not by
  // the handle's address -- glibc hands a freed model's address to the next model of the same size
  uint64_t uid = 0;
  wb_dims dims{};
  int device = 0;
  int compute_dtype = WB_F32;
  int ln_eps_inside_sqrt = 0;
  // 0: mel frames per window bounded by n_audio_ctx, as the reference asserts (mod.rs:236-241);
  // 1: ENCODER POSITIONS bounded by n_audio_ctx = 2 n_audio_ctx frames (Whisper's own 30 s geometry; opt-in)
  int frame_limit_x2 = 0;
  int max_mel_frames() const { return frame_limit_x2 ? 2 * dims.n_audio_ctx : dims.n_audio_ctx; }
  // all weights live in one arena allocation
  wb::DevMem arena;
  wb::DevMem arena_split;     // exact-f32 models with the split-precision encoder: fp16 hi / lo copies of the encoder-side weights
  // range guard of the split-precision kernel: a mapped host word the kernel raises when a result is not finite; once it
  // has tripped the model stays on the exact-f32 kernel (split_off)
  // The model's own flag word serves the stateless entry points only (api.cpp serialises them); every session carries
  // its own words (session.h: guard_host), so guarded passes of one model on different streams share nothing but
  // split_off -- the one field of a loaded model that ever changes (0 -> 1, once); atomic, read once per GEMM dispatch.
  int* split_flag_host = nullptr; int* split_flag_dev = nullptr; int split_off = 0;
  bool split_active() const { return arena_split.p && !__atomic_load_n(&split_off, __ATOMIC_ACQUIRE); }
  // decoder side (batch-mode skinny GEMM on fp16 hi / lo tiles): its own arena and off switch; the flag words are the
  // sessions' (guard_host[1]), checked wherever a decode synchronises with the host (session.cpp: dec_split_check): a trip
  // fails THAT session's call loudly and switches the model to the exact-f32 decoder GEMMs, so the caller's retry succeeds;
  // other sessions drop their captured step graphs the next time they look one up (the graph key carries the switch).
  wb::DevMem arena_dec_split;
  int dec_split_off = 0;
  bool dec_split_active() const { return arena_dec_split.p && !__atomic_load_n(&dec_split_off, __ATOMIC_ACQUIRE); }
  ~wb_model() {
    if (split_flag_host) (void)hipHostFree(split_flag_host);
  }
  // encoder
  wb::LinearW conv1;   // repacked [240 = ci*3+kk][d]
  wb::LinearW conv2;   // repacked [3d = kk*d+ci][d]
  float* enc_pos = nullptr;   // [n_audio_ctx][d]
  std::vector<wb::EncBlockW> enc;
  wb::LayerNormW ln_post;
  // decoder
  float* tok_emb = nullptr;    // E   [V][d]
  float* tok_emb_t = nullptr;  // E^T [d][vocab_ld]  (logits GEMM/GEMV streams V contiguously)
  int vocab_ld = 0;            // V rounded up to 64
  float* dec_pos = nullptr;    // [n_text_ctx][d]
  std::vector<wb::DecBlockW> dec;
  wb::LinearW ckv_all;         // cross K|V projections of ALL layers: [d][n_layer*2d]
  wb::LayerNormW ln_dec;
  float qk_scale = 0.f;        // (d/H)^-0.25 as f32, mod.rs:503
  hipStream_t stream = nullptr;
  // scratch of the stateless entry points (grow-only; serialised by a process-wide lock in api.cpp)
  wb::Workspace ws;
  wb::DevMem io_a, io_b, io_c;
};
