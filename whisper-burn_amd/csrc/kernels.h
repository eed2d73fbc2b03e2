// Host-callable launchers of the gfx950 kernels (defined in the .hip files).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

namespace wb {

// ---- per-kernel profiling (wb_profile_enable): the call site tags the NEXT launch of this thread with a kernel
// class and its algorithmic bytes; the launcher hands the tag's start / stop events to the dispatch itself
// (hipExtLaunchKernelGGL), so the elapsed time is that kernel's own begin -> end -- the quantity
// `rocprofv3 --kernel-trace` reports.  With profiling off a tag costs one predictable branch.
enum KernelClass {
  KC_PREPARE = 0, KC_ATTN_FUSED, KC_CROSS_ATTN, KC_GEMV_COUT, KC_MLP_FUSED, KC_LOGITS, KC_TOPK_MERGE,
  KC_GEMV_LN_QKV, KC_SELF_ATTN, KC_GEMV_OUT, KC_GEMV_LN_CQ, KC_GEMV_LN_MLP1, KC_GEMV_MLP2, KC_CROSS_FUSED,
  // batch mode (more than 8 live rows): one class per kernel of the per-layer chain
  KC_B_RESOLVE_LN, KC_B_GEMM, KC_B_SELF_ATTN, KC_B_CROSS_STREAM, KC_B_CROSS_CHUNK, KC_B_COMBINE, KC_B_GELU_FOLD,
  KC_B_LOGITS_GEMM, KC_B_TOPK_ROWS, KC_PERSIST, KC_BEAM_UPDATE, KC_FOLD_LN_ROWS,
  KC_COUNT
};
void prof_tag(int cls, double algo_bytes);
bool prof_take_events(hipEvent_t* start, hipEvent_t* stop);
#define WB_KLAUNCH(kernel, grid, block, shmem, stream, ...)                                        \
  do {                                                                                             \
    hipEvent_t _pa, _pb;                                                                           \
    if (wb::prof_take_events(&_pa, &_pb))                                                          \
      hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, _pa, _pb, 0, __VA_ARGS__);         \
    else                                                                                           \
      hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                         \
  } while (0)


// ---- mel frontend (mel.hip) ---------------------------------------------------------
constexpr int MEL_N_FFT = 400, MEL_HOP = 160, MEL_N_MELS = 80, MEL_N_BINS = 201, MEL_MAX_TAPS = 16;

struct MelTables {            // device-resident constants, built once per (device, sample_rate)
  float hann[400];            // audio.rs:272-278
  float2 tw[400];             // W_400^{n2*k1}, index n2*20+k1
  int tap_start[80];          // sparse Slaney filterbank rows (audio.rs:67-143)
  int tap_len[80];
  float tap_w[80 * MEL_MAX_TAPS];
};
// Host-side construction (f32 op order of the reference); returns 0 / -1 if a row has > MAX_TAPS taps.
int mel_tables_build(double sample_rate, MelTables* host_out);

struct MelWindow {            // one window of PCM
  int64_t pcm_off;            // offset (samples) of the window start in the PCM buffer
  int32_t n_samples;          // window length in samples (>= 400)
  int32_t n_frames;           // n_samples / 160: frames that enter the max (audio.rs:41-42, :50)
  int32_t n_emit;             // frames written: min(n_frames, clip)  (transcribe.rs:171-177 clips AFTER prep_audio)
  int32_t reserved;
};

// (log10(max(mel,1e-10)) + 4) / 4 for every (window, mel row, frame), zeros for the frames [n_emit, n_emit + pad) below
// pad_limit, + per-tile (max, min) of the un-normalised values.
// out[w][m][t] at out + w*win_stride + m*row_stride + t.  bmax: n_windows * mel_bmax_stride(max_frames) * 2 floats.
int mel_bmax_stride(int max_frames);
void launch_mel_spectrogram(hipStream_t st, const float* pcm, const MelWindow* wins_dev, int n_windows,
                            int max_frames, const MelTables* tabs_dev, float* out, int64_t win_stride,
                            int row_stride, float* bmax_dev, int pad, int pad_limit);
// clamp fix-up: max(x, gmax - 8) (audio.rs:52) on the tiles that need it (normally none: the log-mel is written once)
void launch_mel_finalize(hipStream_t st, const MelWindow* wins_dev, int n_windows, float* out, int64_t win_stride,
                         int row_stride, const float* bmax_dev, int max_frames);
void launch_fill_f32(hipStream_t st, float* p, int64_t n, float v);
// 16-bit PCM -> f32 with the reference's scale s / 32767 (bin/transcribe/main.rs:45-52), correctly rounded division
void launch_pcm_s16_to_f32(hipStream_t st, const int16_t* src, int64_t n, float* dst);

// ---- GEMM (gemm.hip): C = act(A*B + bias) (+ residual) (+ aux[aux_idx[m]]) --------------
enum { ACT_NONE = 0, ACT_GELU = 1 };
struct RowDesc {             // optional per-row A addressing (implicit-GEMM conv, packed windows)
  int32_t off;               // element offset of the row start (may be negative: masked k never read)
  int16_t klo, khi;          // A[m][k] := 0 for k outside [klo, khi)   (ROWS mode)
                             // CONV1 mode: klo = 1 if first frame of its window, khi = 1 if last
};
struct GemmArgs {
  const float* A = nullptr; int64_t lda = 0;     // ROWS: A[m][k] = A[m*lda + k] (or A[desc[m].off + k])
  const RowDesc* a_desc = nullptr;               // per-row descriptors (device), or null
  int a_mask_align = 1;                          // the caller's guarantee: every klo / khi of a_desc is a multiple of this
                                                 // (the split-precision kernel stages whole octets: it needs a multiple of 8)
  int conv1_tstride = 0;                         // >0 selects CONV1 gather: A[m][ci*3+kk] = A[off + ci*tstride + kk - 1]
  const float* B = nullptr; int ldb = 0;         // [K][N] row-major
  float* C = nullptr; int ldc = 0;
  const float* bias = nullptr;                   // [N]
  const float* residual = nullptr; int ldr = 0;  // [M][N], added after activation.  May alias C ONLY with ksplit <= 1: every
                                                 // kernel reads residual[row][col] and writes C[row][col] from the same thread, once
  const float* aux = nullptr; const int32_t* aux_idx = nullptr; int ld_aux = 0;  // + aux[aux_idx[m]][n]
  float col_scale = 1.f; int col_scale_period = 0, col_scale_width = 0;  // out *= col_scale where (n % period) < width
  int M = 0, N = 0, K = 0;                       // K % 16 == 0
  int act = ACT_NONE;
  int ksplit = 1; int64_t c_split_stride = 0;    // split-K: raw partials of K-slice z go to C + z * c_split_stride
  // column blocks: output column n lands at C + (n / c_block_cols) * c_block_stride + row * ldc + n % c_block_cols (0: plain).
  // The cross-K/V projection of ALL decoder layers is one GEMM over [d][n_layer * 2d]; its output is laid out layer-major
  // ([layer][row][2d]) so that a decode step's pass over ONE layer's cached K/V is one dense stream (session.cpp).
  int c_block_cols = 0; int64_t c_block_stride = 0;
  int* range_flag = nullptr;                     // split-precision kernel only: set to 1 (system scope) when a result is not
                                                 // finite, i.e. an operand left fp16's range (|x| >= 65504)
  // split-precision kernel only -- activations as fp16 PIECES (hi = fp16(x), lo = fp16((x - hi) 2^11); two [M][ld] planes,
  // the same 4 bytes per element as f32), written ONCE by the producer instead of being split by every column block of every
  // consumer GEMM on its way into LDS (round 5: ~100 of the ~226 non-MFMA instructions per k-tile of that kernel):
  const uint16_t* Ah = nullptr; const uint16_t* Al = nullptr;   // A pre-split: A[m][k] pieces at [m * lda + k] (plain rows only)
  uint16_t* Ch = nullptr; uint16_t* Cl = nullptr;               // C as pieces at [row * ldc + col] INSTEAD of the f32 C
};
int launch_gemm_f32(hipStream_t st, const GemmArgs& a);   // exact-f32 MFMA (v_mfma_f32_32x32x2_f32)
// split precision (gemm_f16x3.hip): f32-grade products from three fp16 MFMAs; weight pre-split as Wh, Wl [N][ldwt] fp16
int launch_gemm_f16x3(hipStream_t st, const GemmArgs& a, const uint16_t* Wh, const uint16_t* Wl, int ldwt);
void launch_split_weight_f16(hipStream_t st, const float* W, int K, int N, uint16_t* hi, uint16_t* lo);

// ---- elementwise / normalisation (ops.hip) ----------------------------------------
// y[m] = LN(x[m]) * g + b, biased variance; eps placement per variant (see whisper_hip.h)
void launch_layernorm(hipStream_t st, const float* x, float* y, int M, int d, const float* g, const float* b,
                      float eps, int eps_inside_sqrt);
// ... with the result as fp16 pieces (GemmArgs::Ah / Al): yh, yl [M][d]
void launch_layernorm_pieces(hipStream_t st, const float* x, uint16_t* yh, uint16_t* yl, int M, int d, const float* g,
                             const float* b, float eps, int eps_inside_sqrt);
// x[r] = E[tok[r]] + pos[r % L]    (mod.rs:141-146)
void launch_embed(hipStream_t st, const int32_t* tok, int n_rows, int L, int d, const float* E, const float* pos,
                  float* x);

// ---- attention (attention.hip) ---------------------------------------------------
struct AttnSeg { int32_t q_row0, q_len, kv_row0, kv_len; };   // one (batch) segment of packed rows
// O[q][h*64..] = softmax((Q*s)(K*s)^T [+causal]) V per segment and head, head size 64 (mod.rs:493-533)
void launch_attention_f32(hipStream_t st, const float* Q, int ldq, const float* K, const float* V, int ldkv,
                          float* O, int ldo, const AttnSeg* segs_dev, int n_segs, int max_q_len, int n_head,
                          float scale, int causal);
// the same with the choice of arithmetic: split = the 16-bit matrix path (fp16 hi / lo operands, three MFMAs per product, f32
// accumulate: f32-grade; operands must stay inside fp16's range -- see attention.hip), else exact f32
// Oh / Ol non-null: the 16-bit kernel writes its output as fp16 pieces there (ldo halves per row) instead of f32 into O.
// Returns true when it did (the caller's next GEMM then reads pieces), false when O holds f32 (exact-f32 kernels).
bool launch_attention(hipStream_t st, const float* Q, int ldq, const float* K, const float* V, int ldkv,
                      float* O, int ldo, const AttnSeg* segs_dev, int n_segs, int max_q_len, int n_head,
                      float scale, int causal, bool split, uint16_t* Oh = nullptr, uint16_t* Ol = nullptr);

}  // namespace wb
