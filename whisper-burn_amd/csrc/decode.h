// Launchers of the KV-cached decode-step kernels (decode.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "wb_internal.h"

namespace wb {

constexpr int MAX_BEAMS = 8;     // live beams per window (reference: beam_size 5, transcribe.rs:232)
constexpr int TOPK_MAX = 8;
constexpr int CA_STRIDE = 66;    // cross-attention chunk partial: m, l, o[64]

// Per-step state block in device memory (ints).  Header, then per-slot arrays, then per-window lists.
enum { ST_N = 0, ST_STEP = 1, ST_HDR = 4 };
struct StepLayout {
  int S = 0, W = 0;                                  // slot capacity, windows
  int tok = 0, parent = 0, len = 0, win = 0;         // offsets of int[S] arrays
  int win_nb = 0, win_slots = 0;                     // int[W], int[W][MAX_BEAMS]
  int total = 0;
};
inline StepLayout make_step_layout(int S, int W) {
  StepLayout l;
  l.S = S; l.W = W;
  l.tok = ST_HDR; l.parent = l.tok + S; l.len = l.parent + S; l.win = l.len + S;
  l.win_nb = l.win + S; l.win_slots = l.win_nb + W; l.total = l.win_slots + W * MAX_BEAMS;
  return l;
}

enum { PRO_PLAIN = 0, PRO_GELU = 1, PRO_ATTN = 2 };
struct GemvArgs {
  const float* W = nullptr; int ldw = 0;    // [K][ldw] row-major
  int K = 0, N = 0, KS = 1, KSL = 0;        // K-split count / slice length
  int pro = PRO_PLAIN;
  const float* src = nullptr; int ld_src = 0;   // PLAIN: [rows][K]; GELU: partials [KSp][S][K]; ATTN: chunk partials
  const float* pbias = nullptr; int KSp = 0;    // GELU prologue
  int n_head = 0, n_chunks = 0;                 // ATTN prologue
  float* P = nullptr;                       // out partials [KS][S][N]
  const int* st = nullptr; int S = 0;
};

void gemv_plan(int K, int N, int* KS, int* KSL);
int cross_attn_chunk();

void launch_dec_prepare(hipStream_t st, const int* state, const StepLayout& lay, int n_max, const int* tab_old,
                        int* tab_new, int Lmax, const float* E, const float* pos, int d, float* x);
void launch_dec_resolve_ln(hipStream_t st, const int* state, int n_max, float* x, const float* P, int KS, int S,
                           const float* bias, int d, const LayerNormW& ln, int eps_inside_sqrt, float* h);
void launch_dec_gemv(hipStream_t st, const GemvArgs& a, int n_rows_hint);
void launch_dec_self_attn(hipStream_t st, const int* state, const StepLayout& lay, int n_max, int n_head,
                          const float* Pqkv, int KS, const float* bqkv, int d, float* Kc, float* Vc, const int* tab,
                          int Lmax, float scale, float* att);
void launch_dec_cross_attn(hipStream_t st, const int* state, const StepLayout& lay, int n_windows, int n_head,
                           int n_chunks, const float* Pq, int KS, const float* bq, int d, const float* ckv, int ldkv,
                           int koff, const int* win_row0, const int* win_C, float scale, float* ca);
void launch_dec_topk(hipStream_t st, const int* state, int n_max, const float* logits, int KS, int64_t plane, int V,
                     const float* mask, int use_mask, int k, int32_t* out_id, float* out_lp, float* row_stats);
void launch_dec_logprob_row(hipStream_t st, const float* x, int KS, int64_t plane, int V, const float* mask,
                            int use_mask, const float* stats, float* out);

}  // namespace wb
