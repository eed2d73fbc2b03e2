// Launchers of the KV-cached decode-step kernels (decode.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <algorithm>

#include "kernels.h"
#include "wb_internal.h"

namespace wb {

constexpr int MAX_BEAMS = 8;     // live beams per window (reference: beam_size 5, transcribe.rs:232)
constexpr int TOPK_MAX = 8;
constexpr int CA_STRIDE = 66;    // cross-attention chunk partial: m, l, o[64]
constexpr int CA_NCH_MAX = 12;   // most key chunks a (row, head) is split into: ceil(1500 / 128)
constexpr int TS_STRIDE = 2 + 2 * TOPK_MAX;   // logits tile partial: max, sum-exp, k x (value, id)
constexpr int GV_CT = 128;       // columns per GEMV block tile
constexpr int GV_CT_LOGITS = 128; // logits tile (64 = twice the blocks measured slower: 17.7 vs 16.3 us, merge 6.3 vs 4.4 us)
constexpr int KS_MAX = 16;       // most K-split partials any consumer folds

// Per-step state block (ints): header, per-slot arrays, per-window lists.  The host writes it into
// mapped pinned memory; dec_prepare_kernel copies it to device memory for the rest of the step.
enum { ST_N = 0, ST_STEP = 1, ST_HDR = 4 };
// Device control block of the chained greedy decode (ints): step, then cur_tok[S], done[S], out_len[S].
enum { GC_STEP = 0, GC_NDONE = 1, GC_ALLDONE = 2, GC_BAD = 3, GC_HDR = 4 };   // [1]: windows finished so far, [2]: all of them,
                                                                          // [3]: a row had no finite log-prob (top1_or_eot)
struct StepLayout {
  int S = 0, W = 0;                                  // slot capacity, windows
  int tok = 0, parent = 0, len = 0, win = 0;         // offsets of int[S] arrays
  int dead = 0;                                      // int[S]: 1 = this row's window already ended (chained greedy decode):
                                                     // its attention blocks exit at once and stream no cached K/V
  int win_nb = 0, win_slots = 0;                     // int[W], int[W][MAX_BEAMS]
  int total = 0;
};
inline StepLayout make_step_layout(int S, int W) {
  StepLayout l;
  l.S = S; l.W = W;
  l.tok = ST_HDR; l.parent = l.tok + S; l.len = l.parent + S; l.win = l.len + S; l.dead = l.win + S;
  l.win_nb = l.dead + S; l.win_slots = l.win_nb + W; l.total = l.win_slots + W * MAX_BEAMS;
  return l;
}

enum { PRO_PLAIN = 0, PRO_GELU = 1, PRO_ATTN = 2, PRO_LN = 3 };
struct GemvArgs {
  const float* W = nullptr; int ldw = 0;    // [K][ldw] row-major
  int K = 0, N = 0, KS = 1, KSL = 0;        // K-split count / slice length
  int pro = PRO_PLAIN;
  // PLAIN: src [rows][K].  GELU: src = partials [KSp][S][K] of lin1, pbias.  ATTN: src = chunk partials.
  // LN: src = residual stream x_in [S][K] (K == d); pending partials pend [KSp][S][K] + pbias are folded
  //     first (KSp = 0: none), block (0,0) writes the folded stream to x_out, then LayerNorm(g, b, eps).
  const float* src = nullptr; int ld_src = 0;
  const float* pend = nullptr; const float* pbias = nullptr; int KSp = 0;
  float* x_out = nullptr;
  const float* ln_g = nullptr; const float* ln_b = nullptr; float ln_eps = 0.f; int ln_inside = 0;
  int n_head = 0, n_chunks = 0;             // ATTN prologue
  float* P = nullptr;                       // out partials [KS][S][N]
  float* h_tmp = nullptr;                   // [S][K] scratch: LN rows of the 9 - 16-row logits pass (dec_fold_ln_rows_kernel)
  // STATS epilogue (logits): + mask, per-tile max / sum-exp / top-k -> tstats [row][n_tiles][TS_STRIDE]
  const float* mask = nullptr; int use_mask = 0, topk = 0; float* tstats = nullptr;
  const int* st = nullptr; int S = 0;
  int ct = 0;                               // columns per block (0 = 128); multiple of 4
};

void gemv_plan(int K, int N, int* KS, int* KSL);
int cross_attn_chunk();

void launch_dec_prepare(hipStream_t st, const int* state_host_mapped, int* state_dev, const StepLayout& lay, int n,
                        int* tabs, int Lmax, const float* E, const float* pos, int d, float* x, const int* gctl);
void launch_dec_resolve_ln(hipStream_t st, const int* state, int n_max, const float* x_in, float* x_out,
                           const float* P, int KS, int S, const float* bias, int d, const LayerNormW& ln,
                           int eps_inside_sqrt, float* h);
void launch_dec_gemv(hipStream_t st, const GemvArgs& a, int n_rows_hint, bool stats);
bool dec_logits_two_launches(const GemvArgs& a, int n_rows_hint);   // (a with pro = PRO_LN, KS, K, ct, h_tmp set)
void launch_dec_self_attn(hipStream_t st, const int* state, const StepLayout& lay, int n_max, int n_head,
                          const float* Pqkv, int KS, const float* bqkv, int d, float* Kc, float* Vc, const int* tab,
                          int Lmax, float scale, float* att);
// fused query projection of the cross-attention kernel (small models; see dec_cross_attn_kernel)
struct CaFuse {
  const float* x_in = nullptr;    // residual stream [S][d]
  const float* pend = nullptr; int KSp = 0; const float* pbias = nullptr;   // pending out-projection partials + bias
  float* x_out = nullptr;         // folded residual stream (written by block chunk 0 / head 0 of each window)
  const float* ln_g = nullptr; const float* ln_b = nullptr; float ln_eps = 0.f; int ln_inside = 0;
  const float* Wq = nullptr;      // [d][d] row-major
  // fused out-projection (optional): Wo [d][d]; records {m, l, P[d]} go to rec [n_head * n_chunks][S][2 + d]
  const float* Wo = nullptr;
  float* rec = nullptr;
};
bool cross_attn_can_fuse_q(int d);
void launch_dec_cross_attn(hipStream_t st, const int* state, const StepLayout& lay, int n_windows, int n_head,
                           int n_chunks, const float* Pq, int KS, const float* bq, int d, const float* ckv, int ldkv,
                           int koff, const int* win_row0, const int* win_C, float scale, float* ca, int max_nb,
                           const CaFuse* fuse = nullptr);
// batch mode, one beam per window: one block per (head, window) streams the whole cached K/V and writes the normalised
// head outputs to att [S][d] (no chunk partials, no combine launch)
// the streaming kernel with the front of the sublayer inside (fold of the pending planes, cross_attn_ln, Wq): see decode.hip
constexpr int CSF_MAX_D = 1280;
struct CaStreamFuse {
  const float* x_in = nullptr; float* x_out = nullptr;            // residual stream in / folded stream out (head 0's blocks)
  const float* pend = nullptr; int KSp = 0; const float* pbias = nullptr;   // the self-attention out-projection's planes + bias
  const float* ln_g = nullptr; const float* ln_b = nullptr; float ln_eps = 0.f; int ln_inside = 0;
  const float* Wq = nullptr;                                      // [d][d]
};
bool cross_stream_can_fuse(int d);
void launch_dec_cross_attn_stream_fused(hipStream_t st, const int* state, const StepLayout& lay, int n_windows, int n_head,
                                        const float* bq, int d, const float* ckv, int ldkv, int koff, const int* win_row0,
                                        const int* win_C, float scale, float* att, const CaStreamFuse& fz);
void launch_dec_cross_attn_stream(hipStream_t st, const int* state, const StepLayout& lay, int n_windows, int n_head,
                                  const float* Pq, int KS, const float* bq, int d, const float* ckv, int ldkv, int koff,
                                  const int* win_row0, const int* win_C, float scale, float* att);
// what the merge kernel needs to prepare the NEXT chained step (x == nullptr: it does not)
struct NextPrep {
  float* x = nullptr;        // residual-stream rows [S][d] the next step starts from
  const float* E = nullptr;  // token embedding [V][d]
  const float* pos = nullptr;   // decoder positional embedding [n_text_ctx][d]
  int* tabs = nullptr;       // position tables, double-buffered by step parity
  int d = 0;
  // mapped host memory, int[S][2] = (steps completed, finished) per row: the host polls it instead of copying the
  // control block back and synchronising the stream after every chunk of chained steps
  int* hflags = nullptr;
};
void launch_dec_topk_merge(hipStream_t st, int* state, int n_max, const float* tstats, int n_tiles, int k,
                           int32_t* out_id, float* out_lp, float* row_stats, const StepLayout& lay, int* gctl, int* gtok,
                           int Lmax, int eot, const NextPrep& nx);
void launch_dec_gelu_fold(hipStream_t st, const int* state, int n_max, const float* P, int KS, int S, int K,
                          const float* bias, float* out);
void launch_dec_attn_combine(hipStream_t st, const int* state, int n_max, const float* ca, int n_head, int n_chunks,
                             float* out);
void launch_dec_topk_rows(hipStream_t st, int* state, int n_max, const float* logits, int V, const float* mask,
                          int use_mask, int k, int32_t* out_id, float* out_lp, float* row_stats, const StepLayout& lay,
                          int* gctl, int* gtok, int Lmax, int eot);
void launch_dec_logprob_row(hipStream_t st, const float* x, int KS, int64_t plane, int V, const float* mask,
                            int use_mask, const float* stats, float* out);

// ---- device-chained beam search (src/beam.rs:9-110 + transcribe.rs:253-312 on the device) -------------------------
// The host-driven beam search (transcribe.cpp: beam_search_windows) pays a synchronisation, a PCIe read of the top-k and the
// beam bookkeeping per step (~45 us of a ~240 us step at tiny.en).  Here the bookkeeping is a one-block kernel between two
// steps: it reads the step's top-k rows, applies beam_search_step (beam.rs:39-79) window by window in f64 exactly as the host
// code does, runs the termination test of beam.rs:23-27 and writes the NEXT step's state block (token, parent slot, length,
// window of every live beam) to device memory, where dec_prepare_kernel picks it up: the host enqueues step after step
// without looking.  Sequences are a tree of (token, parent node) pairs -- one node per (step, window, beam) -- walked back
// by the host at the end.
constexpr int BEAM_KB = 2 * TOPK_MAX;     // beams a window can hold: k new + k finished (beam.rs:72-78)
enum { BC_DEPTH = 0, BC_ALLDONE = 1, BC_ERR = 2, BC_NLIVE = 3, BC_HDR = 8 };   // header ints of the control block
struct BeamChainLayout {
  int W = 0, max_depth = 0;
  // int offsets: nb[W], done[W], node / fin / prev_slot / slot_now [W][BEAM_KB]; then doubles lp[W][BEAM_KB] (8-byte aligned);
  // then the node pool int2[(max_depth + 1) * W * BEAM_KB]
  int nb = 0, done = 0, node = 0, fin = 0, prev_slot = 0, slot_now = 0, lp = 0, nodes = 0, total_ints = 0;
};
inline BeamChainLayout make_beam_layout(int W, int max_depth) {
  BeamChainLayout b;
  b.W = W; b.max_depth = max_depth;
  b.nb = BC_HDR; b.done = b.nb + W; b.node = b.done + W; b.fin = b.node + W * BEAM_KB; b.prev_slot = b.fin + W * BEAM_KB;
  b.slot_now = b.prev_slot + W * BEAM_KB;
  b.lp = (b.slot_now + W * BEAM_KB + 1) & ~1;
  b.nodes = b.lp + 2 * W * BEAM_KB;
  b.total_ints = b.nodes + 2 * (max_depth + 1) * W * BEAM_KB;
  return b;
}
struct BeamChainArgs {
  int* ctl = nullptr; BeamChainLayout bl;
  const int32_t* topk_id = nullptr; const float* topk_lp = nullptr;   // [S][TOPK_MAX] of the step just run (device memory)
  int* state_out = nullptr; StepLayout lay;                            // the next step's state block (device memory)
  int k = 0, eot = 0, first = 0;                                       // first: no step has run yet (termination test + slots only)
  int V = 0;                                                           // vocabulary: a top-k id outside [0, V) (NaN logits) ends its beam and raises BC_ERR
  int step_pos = 0;                                                    // position index (ST_STEP) of the first decode step
  int* done_flag_host = nullptr;                                       // mapped host word: set when every window has ended
  // dec_prepare_kernel's work for the next step, done here (one launch less per step): position tables + embedding rows
  int* tabs = nullptr; int Lmax = 0; const float* E = nullptr; const float* pos = nullptr; int d = 0; float* x = nullptr;
};
void launch_dec_beam_update(hipStream_t st, const BeamChainArgs& a);

// ---- fused small-batch sublayer kernels (decode_fused.hip) ----------------------------------------------
// Common prologue of both: x = x_in + (pbias + sum_s pend[s]) (KSp planes of [S][d]; KSp = 0: none), block 0
// writes x to x_out, then LayerNorm(ln_g, ln_b, ln_eps).
struct MlpFusedArgs {
  const int* st = nullptr; int S = 0, d = 0;
  const float* x_in = nullptr; const float* pend = nullptr; int KSp = 0; const float* pbias = nullptr; float* x_out = nullptr;
  // n_chunks > 0: `pend` holds cross-attention chunk records [n_head * n_chunks][S][2 + d] = {m, l, o_chunk Wo} instead of
  // plain partial planes (KSp = n_head * n_chunks): the fold weights plane (h, c) by exp(m_hc - M_h) / sum_c exp(..) l_hc
  int n_head = 0, n_chunks = 0;
  const float* ln_g = nullptr; const float* ln_b = nullptr; float ln_eps = 0.f; int ln_inside = 0;
  const float* W1 = nullptr; int ld1 = 0; const float* b1 = nullptr;     // [d][4d], [4d]
  const float* W2 = nullptr;                                             // [4d][d]
  float* P = nullptr;                                                    // out: planes [4d / 64][S][d] (lin2 bias NOT added)
  unsigned long long* stamps = nullptr;                                  // developer probe (-DWB_STAMPS): phase clock of block 0
  // row groups (9 - 16 live rows on the fused sublayer path): group g = blockIdx.y handles rows [8 g, 8 g + 8); the kernel
  // wrapper sets row0 and shifts x_in / pend / x_out / P by row0 rows (plain planes only, not the record mode)
  int row0 = 0;
  // persistent mode only: the same streams / planes as 8-byte {tag, value} granules (handoff.h)
  const void* g_x_in = nullptr; const void* g_pend = nullptr; void* g_x_out = nullptr; void* g_P = nullptr;
};
struct AttnFusedArgs {
  const int* st = nullptr; StepLayout lay; int S = 0, d = 0, n_head = 0;
  const float* x_in = nullptr; const float* pend = nullptr; int KSp = 0; const float* pbias = nullptr; float* x_out = nullptr;
  const float* ln_g = nullptr; const float* ln_b = nullptr; float ln_eps = 0.f; int ln_inside = 0;
  const float* Wqkv = nullptr; int ldqkv = 0; const float* bqkv = nullptr; float scale = 0.f;   // [d][3d], [3d]
  float* Kc = nullptr; float* Vc = nullptr; const int* tabs = nullptr; int Lmax = 0;            // this layer's cache
  const float* Wo = nullptr;                                                                    // [d][d]
  float* P = nullptr;                                                    // out: planes [n_head][S][d] (out bias NOT added)
  unsigned long long* stamps = nullptr;                                  // developer probe (-DWB_STAMPS): phase clock of block 0
  const void* g_x_in = nullptr; const void* g_pend = nullptr; void* g_x_out = nullptr; void* g_P = nullptr;   // persistent mode
};
struct CrossFusedArgs {
  const int* st = nullptr; StepLayout lay; int S = 0, d = 0, n_head = 0;
  const float* x_in = nullptr; const float* pend = nullptr; int KSp = 0; const float* pbias = nullptr; float* x_out = nullptr;
  const float* ln_g = nullptr; const float* ln_b = nullptr; float ln_eps = 0.f; int ln_inside = 0;
  const float* Wq = nullptr; const float* bq = nullptr; float scale = 0.f;         // [d][d], [d]
  const float* ckv = nullptr; int ldkv = 0, koff = 0;                             // cached cross K|V rows (K pre-scaled)
  const int* win_row0 = nullptr; const int* win_C = nullptr;
  int n_pass = 1;                                                                 // 2: some window has CROSS_FUSED_MAX_C < C <= 2 CROSS_FUSED_MAX_C keys
  const float* Wo = nullptr;                                                      // [d][d]
  float* P = nullptr;                                                             // out: planes [n_head][S][d] (out bias NOT added)
  unsigned long long* stamps = nullptr;
  const void* g_x_in = nullptr; const void* g_pend = nullptr; void* g_x_out = nullptr; void* g_P = nullptr;   // persistent mode
};
// ---- batch mode: skinny split-K weight-stream GEMM (decode_batch.hip) ---------------------------------------------
struct SkinnyArgs {
  const float* A = nullptr; int lda = 0;          // activations [M][K]
  const float* B = nullptr; int ldb = 0;          // weight [K][N] row-major (Linear: [d_in, d_out])
  int M = 0, N = 0, K = 0, ksplit = 1;            // M <= 64, N % 64 == 0, K % (32 ksplit) == 0
  float* P = nullptr; int plane = 0;              // split-K planes out: P[z * plane + r * N + n]  (plane in elements)
  // split-precision variant: the weight as fp16 hi / lo (x 2^11) pieces in MFMA tiles (LinearW::th / tl); null = exact f32
  const uint16_t* Bh = nullptr; const uint16_t* Bl = nullptr;
  int* range_flag = nullptr;                      // raised (mapped host word) when a result of a LIVE row is not finite
  const int* st = nullptr;                        // step state (st[ST_N] = live rows: rows past it hold stale activations); null: all M
};
// W [K][N] f32 -> hi, lo tiles [N / 16][K / 32][4][16][8] fp16 (K % 32 == 0, N % 16 == 0)
void launch_split_weight_f16_tiles(hipStream_t st, const float* W, int K, int N, uint16_t* hi, uint16_t* lo);
int skinny_ksplit(int K, int N, int max_ks, int max_rows);   // 0: shape not served
bool skinny_supported(int M, int K, int N);
int launch_dec_skinny_gemm(hipStream_t st, const SkinnyArgs& a);

// ---- persistent flag-chained greedy decode (decode_persist.hip) --------------------------------------------------
// ONE co-resident grid runs every sublayer of every step of the device-chained greedy loop: a block executes the
// roles i = blockIdx.x, + gridDim.x, ... of a per-step role list in dependency order (self-attention block (head, row),
// cross-attention block (head, row), MLP block (64 hidden units), logits tile (128 vocabulary columns), merge (row)),
// streams each role's weights BEFORE it waits for the arrival counter of the role's producers, and publishes its output
// planes with write-through stores (handoff.h).  No kernel boundary, no grid barrier, fixed summation orders.
struct PsLayerArgs { AttnFusedArgs attn; CrossFusedArgs cross; MlpFusedArgs mlp; };
enum { PSR_ATTN = 0, PSR_CROSS = 1, PSR_MLP = 2, PSR_LOGITS = 3, PSR_MERGE = 4, PSR_FINLN = 5 };
struct PsRole { int kind, layer, a, b; };     // a: head / hidden slice / first tile, b: row (logits: tiles of the role; layer: its ordinal)
constexpr int PS_MAX_FORCED = 8;
struct PersistArgs {
  const PsLayerArgs* layers = nullptr;        // [n_layer] (device)
  const PsRole* roles = nullptr; int n_roles = 0;   // one step's roles, grouped by block, each block's in dependency order (device)
  const int* role_off = nullptr;              // [grid + 1]: block b runs roles [role_off[b], role_off[b + 1]) every step
  int n_logits_roles = 0;
  int n_layer = 0, n_rows = 0, S = 0, d = 0, n_head = 0, nb_mlp = 0;
  int n_pass = 1;                             // key passes of the cross-attention roles (2: a window with > CROSS_FUSED_MAX_C keys)
  int* ctl = nullptr;                         // HX_* control words + arrival counters (device; set up by the host)
  int step0 = 0, n_steps = 0;                 // first decode step of the chain, most steps to run
  // prompt prefill inside the launch: the first n_forced steps do not choose their next token -- position step0 + e + 1 holds
  // forced[e] (the rest of the prompt, transcribe.rs:203) -- and skip the logits work
  int n_forced = 0; int forced[PS_MAX_FORCED] = {0, 0, 0, 0, 0, 0, 0, 0};
  int mask_until_len = 0;                     // special-token mask while len <= this (transcribe.rs:271-275)
  // logits role: LN(x_fin + b2 + sum P2) . E^T tile -> (best value, best id) per row and tile (x_fin, P2: granules)
  const void* x_fin = nullptr; const void* P2 = nullptr; const float* b2_last = nullptr;
  void* g_xn = nullptr;                       // [S][d] granules: ln(x + last MLP), written once per row by the final-LN role
  unsigned tag_base = 0;                      // granule tags of this launch: tag_base + 2 + 3 (e n_layer + l) + sublayer
  const float* ln_g = nullptr; const float* ln_b = nullptr; float ln_eps = 0.f; int ln_inside = 0;
  const float* Et = nullptr; int vocab_ld = 0, V = 0; const float* mask = nullptr;
  float* tstats = nullptr; int n_tiles = 0;   // [S][n_tiles][2]
  // merge role: argmax over the tiles, chain bookkeeping, next step's embedding
  int* gctl = nullptr; int* gtok = nullptr; int Lmax = 0, eot = 0;
  const float* E = nullptr; const float* pos = nullptr; void* x0 = nullptr; int* tabs = nullptr;   // x0: granules
  int* dead = nullptr;                        // [S]: rows whose window has ended
  unsigned long long* stamps = nullptr;       // optional timeline: [n_steps][n_roles][3] (role start, wait passed, done)
};
int ps_ctl_ints(int S, int n_layer);
bool dec_persist_supported(int d, int n_rows, int max_keys);
// grid: blocks of 512 threads that are co-resident on this device for (d, n_rows) -- 0 if the kernel cannot run
int dec_persist_max_grid(int device, int d, int n_rows, int max_keys);
int launch_dec_persist(hipStream_t st, const PersistArgs& a, int grid);
// seeds the granule copy of the first step's x rows: tag = tag_base + 1 (what the first self-attention blocks expect)
void launch_ps_seed(hipStream_t st, const float* x, int n, void* gx, unsigned tag);

#if defined(HIPEMU) && !defined(HIPEMU_PROD_GEOMETRY)
// (functional-model build: three tiles per pass, so that micro models run both rings -- one pass at n_audio_ctx = 400
// (C = 200 keys), two passes for its doubled windows (C = 395) and for reference-length windows (C = 745).  The second
// functional-model library, `make prod` in tools/hipemu, compiles the PRODUCT's ring of 768 keys: tests/test_emu_functional.py
// runs the real window lengths through it)
constexpr int CROSS_FUSED_MAX_C = 384;
#else
constexpr int CROSS_FUSED_MAX_C = 768;   // keys per window one pass of the fused cross-attention block holds (n_audio_ctx / 2 = 750)
#endif
constexpr int CROSS_FUSED_MAX_PASSES = 2;   // ... and the passes it can make (C = 1500: the opt-in 30 s window)
void launch_dec_cross_fused(hipStream_t st, const CrossFusedArgs& a, int n_rows_hint);
bool dec_fused_supported(int d);
int dec_mlp_fused_planes(int d);
void launch_dec_mlp_fused(hipStream_t st, const MlpFusedArgs& a, int n_rows_hint);
void launch_dec_attn_fused(hipStream_t st, const AttnFusedArgs& a, int n_rows_hint);

}  // namespace wb
