// Persistent, flag-chained greedy decode: ONE co-resident grid runs every sublayer of every decode step.
//
// Why: with a handful of live rows a decode step is a chain of 3 n_layer + 2 dependent sublayers, each of which streams
// 0.2-0.6 MB of weights / cached K,V through ONE compute unit.  As separate launches (decode_fused.hip) every link pays its
// dispatch, a memory round trip for its input planes, and -- the largest term -- the time a single CU needs to pull the
// block's operands (~5 us of a ~10 us launch), none of which depends on the predecessor.  Here every role of a step is
// resident at once: a block issues the loads of its weights (and, for cross-attention, of the window's whole cached K)
// FIRST, then waits for the arrival counter of its producers, then folds their planes; the operand pull of sublayer
// k + 1 runs under sublayer k.  The reference recomputes the whole decoder per token (/root/reference/src/transcribe.rs:
// 253-307, src/model/mod.rs:345-350); the arithmetic of a step here is the arithmetic of the fused sublayer kernels
// (decode_fused_bodies.h: same device templates, same fixed summation orders -> bit-reproducible).
//
// Roles of one step, in dependency order:
//   per layer l: attn (head h, row r) x H R  ->  cross (h, r) x H R  ->  mlp (64-unit hidden slice j) x 4 d / 64
//   final LN (row r) x R: ln(x + last MLP planes), once per row   ->   logits (a run of 128-column vocabulary tiles)
//   ->   merge (row r) x R
// The host deals the roles to the blocks (session.cpp): a block runs its own list, in dependency order, every step.  The
// first sublayers' blocks take no logits work (they are back at their wait, weights requested, before the step ends).
// Dependencies = arrival counters (handoff.h), monotonic within the launch:
//   attn(0,.,r)  waits for merge(r) of the previous step (its x row)          c_x[r]       >= e
//   attn(l,.,r)  waits for every MLP slice of layer l - 1                      c_mlp[l-1]   >= (e + 1) NB
//   cross(l,.,r) waits for the H attention blocks of its row                   c_attn[l][r] >= (e + 1) H
//   mlp(l,.)     waits for every cross-attention block of the layer            c_cross[l]   >= (e + 1) H R
//   logits(.)    waits for every MLP slice of the last layer                   c_mlp[NL-1]  >= (e + 1) NB
//   merge(r)     waits for every logits role (8 counters, role q arrives at q mod 8)  c_log[k] >= (e + 1) n_k
// Every block's list follows one global dependency order and all blocks are co-resident (the host sizes the grid from
// the occupancy query), so the smallest unfinished role can always run: no deadlock.  Reuse of the plane
// buffers across steps is ordered by the same chain (a role arrives only after its last read).
//
// End of the decode: the merge role of the last window to end stores HX_STOP = first step that must not run; every wait
// polls that word next to its counter, and a block that sees it leaves the kernel (without arriving).  Rows whose window
// has ended are marked dead: their attention roles skip the work and arrive at once.
#include <hip/hip_runtime.h>

#include <climits>
#include <vector>

#include "decode.h"
#include "decode_fused_bodies.h"
#include "handoff.h"
#include "wave_ops.h"

namespace wb {
namespace {

using namespace fused;

constexpr int PS_NT = 512;
constexpr int PS_CT = 128;      // vocabulary columns per logits tile
constexpr int PS_NGO = 16;      // wake-up words of the logits roles (a few hundred pollers: ~13 per word)

// ---- final LayerNorm role: row r of  ln(x + b2 + sum_j P2[j])  (mod.rs:155), once per row ---------------------------------
// A few hundred logits blocks each folding the 4 d / 64 planes themselves pulled 22 MB through the fabric per step (write-
// through data is not shared through the L2s); one block per row folds and normalises, and the logits blocks read d values.
template <int DPL>
__device__ __forceinline__ bool ps_finln_role(const PersistArgs& a, const int r, const PsStep& ps) {
  constexpr int d = 64 * DPL, FP = 4 * DPL;
  __shared__ __attribute__((aligned(16))) float hs[d];
  const int tid = role_tid<true>(), lane = tid & 63, wave = tid >> 6;
  const int c = tid < d ? tid : 0;
  const float g_own = a.ln_g[c], b_own = a.ln_b[c];   // thread = column: every wave normalises its own 64 columns
  float accp = a.b2_last[c];
  if (!ps_wait(ps)) return false;                   // pre-wake: the last layer's cross-attention blocks have finished
  if (ld_i<true>(ps.dead + r) != 0) return true;
  ps_nap<2>();
  const Buf16 xgb(a.x_fin), pgb(a.P2);
  const uint32_t pvo = (uint32_t)(r * d + c);
  const int iplane = a.S * d;
  float v = 0.f;
  unsigned sweeps = 0;
  Gran gx{0u, 0.f}, g[FP];
#pragma unroll
  for (int j = 0; j < FP; j++) g[j] = Gran{0u, 0.f};
  for (;;) {
    if (gx.tag != ps.tag_in) gx = ld_gran(xgb, pvo, 0u);
#pragma unroll
    for (int j = 0; j < FP; j++)
      if (j < a.nb_mlp && g[j].tag != ps.tag_in) g[j] = ld_gran(pgb, pvo, (uint32_t)(j * iplane));
    bool ok = gx.tag == ps.tag_in;
#pragma unroll
    for (int j = 0; j < FP; j++) ok &= (j >= a.nb_mlp) || g[j].tag == ps.tag_in;
    if (ok) {
#pragma unroll
      for (int j = 0; j < FP; j++) accp += (j < a.nb_mlp) ? g[j].v : 0.f;             // s ascending (mod.rs:346-348)
      v = gx.v + accp;
      break;
    }
    if (ps_sweep_retry(sweeps, ps)) { *ps.lds_flag = 0; break; }
  }
  if (tid < d) hs[tid] = v;
  ps_stamp(ps, 2);
  __syncthreads();
  if (!ps_sweeps_ok(ps)) return false;
  if (wave < DPL) {                                 // (statistics per wave: no second barrier, no LDS round trip of the result)
    float mean, denom;
    ln_stats_lds<DPL>(hs, d, lane, a.ln_eps, a.ln_inside, mean, denom);
    const Buf16 xo(a.g_xn);
    st_gran(xo, (uint32_t)(r * d + tid), ps.tag_out, (v - mean) / denom * g_own + b_own);
  }
  return true;
}

// ---- logits role: tiles of  xn . E^T  (mod.rs:156), last position only ----------------------------------------------------
// Both E^T tiles of the role (d x 128 floats = d / 16 float4 per thread each) are in flight before the wait.  The normalised
// rows arrive as granules from the final-LN roles.  Per row and tile the block leaves the best masked logit and its id --
// greedy needs the argmax only (log_softmax is monotone: transcribe.rs:276 with beam.rs k = 1).
template <int MR, int DPL>
__device__ __forceinline__ bool ps_logits_role(const PersistArgs& a, const int tile0, const int n_t, const PsStep& ps,
                                               const bool forced) {
  constexpr int d = 64 * DPL, NT = PS_NT, CT = PS_CT;
  constexpr int NR = d / 16;                        // E^T rows per thread: k = (2 wave + hh) NR + i
  constexpr int EPT = (MR * d + NT - 1) / NT;
  constexpr int PCH = EPT <= 3 ? 8 : EPT <= 6 ? 4 : 2;
  constexpr int NQ = MR * CT / NT;                  // tile values per thread in the column-sum pass
  constexpr bool TWO = DPL <= 6;                    // two tiles resident (2 x d / 16 float4 per thread) where they fit
  __shared__ __attribute__((aligned(16))) float xs[MR][d];
  __shared__ __attribute__((aligned(16))) float red[8][MR][CT];
  __shared__ float tilev[MR][CT];
  const int tid = role_tid<true>(), lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, c4 = (lane & 31) * 4;
  // (buffer loads: one lane offset, the row offset i vocab_ld rides in the scalar offset -- 24 64-bit addresses per lane
  // would otherwise live across the tile loop)
  const Buf16 etb(a.Et);
  auto load_tile = [&](float4 (&w)[NR], int tile) {
    const int n0 = tile * CT;
    const bool col_ok = n0 + c4 < a.vocab_ld;      // lanes past the padded vocabulary re-read the tile's first columns
    const uint32_t vo = (uint32_t)((2 * wave + hh) * NR * a.vocab_ld + n0 + (col_ok ? c4 : 0));
#pragma unroll
    for (int i = 0; i < NR; i++) w[i] = ld_f4_plain(etb, vo, (uint32_t)(i * a.vocab_ld));
  };
  // Both tiles of the role are requested BEFORE the wait where the registers allow it: streamed after the wait, the
  // second tile of 203 blocks (40 MB at once, next to everybody's weight prefetch for the next step) took ~20 us
  // (profiles/r03_c_ps_timeline_phases.txt).
  float4 w0[NR], w1[TWO ? NR : 1];
  if (!forced) {                                    // (a prompt step chooses nothing: no E^T stream)
#if !defined(WB_PS_DRY)
    load_tile(w0, tile0);
    if constexpr (TWO) { if (n_t > 1) load_tile(w1, tile0 + 1); }
#endif
  }
  const int use_mask = (ps.step + 1 <= a.mask_until_len) ? 1 : 0;          // transcribe.rs:271-275
  int deadm = 0;                                    // bit r: row r takes no part (its window has ended)
  if (!ps_wait(ps)) return false;                   // pre-wake: the last layer's cross-attention blocks have finished
  // (no nap: the wake-up word reaches a few hundred pollers ~9 us after the broadcast -- the last MLP has finished by then)
  {
    // the normalised rows: MR d granules, re-read until every live row carries the final-LN roles' tag
    int off[EPT];
#pragma unroll
    for (int i = 0; i < EPT; i++) {
      int e = tid + NT * i;
      if (e >= MR * d) e = tid;
      off[i] = e;
    }
#pragma unroll
    for (int r = 0; r < MR; r++)
      if (r >= ps.n_rows || ld_i<true>(ps.dead + min(r, ps.n_rows - 1)) != 0) deadm |= 1 << r;
    const Buf16 xgb(a.g_xn);
    float v[EPT];
    unsigned sweeps = 0;
    for (;;) {
      Gran gx[EPT];
#pragma unroll
      for (int i = 0; i < EPT; i++) gx[i] = ld_gran(xgb, (uint32_t)off[i], 0u);
      bool ok = true;
#pragma unroll
      for (int i = 0; i < EPT; i++) ok &= ((deadm >> (off[i] / d)) & 1) != 0 || gx[i].tag == ps.tag_in;
      if (ok) {
#pragma unroll
        for (int i = 0; i < EPT; i++) v[i] = ((deadm >> (off[i] / d)) & 1) ? 0.f : gx[i].v;
        break;
      }
      if (ps_sweep_retry(sweeps, ps)) { *ps.lds_flag = 0; break; }
    }
#pragma unroll
    for (int i = 0; i < EPT; i++) {
      const int e = tid + NT * i;
      if (e < MR * d) (&xs[0][0])[e] = v[i];
    }
  }
  ps_stamp(ps, 2);
  __syncthreads();
  if (!ps_sweeps_ok(ps)) return false;
  ps_stamp(ps, 3);
  // a prompt step: the role has kept its place in the chain (it arrives after the final-LN roles, the merge role after it --
  // the order the reuse of the x rows and planes relies on) and is done
  if (forced) return true;
#if defined(WB_PS_DRY)
  // dry build (tools/build_exp.sh dry): the role's tile records without the E^T stream and the GEMV -- token 0-ish every step
  for (int t = 0; t < n_t; t++)
    if (wave < MR && !((deadm >> wave) & 1) && lane == 0) {
      float* ts = a.tstats + ((int64_t)wave * a.n_tiles + tile0 + t) * 2;
      st_f<true>(ts, 0.f);
      st_f<true>(ts + 1, __int_as_float((tile0 + t) * CT));
    }
  return true;
#endif
  // one tile: GEMV from the registers, column sums over the eight waves, mask, per-row best (value desc, id asc)
  auto run_tile = [&](const float4 (&w)[NR], int tile) {
    const int n0 = tile * CT;
    float mk[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int col = n0 + ((tid + NT * q) & (CT - 1));
      mk[q] = (use_mask && col < a.V) ? a.mask[col] : 0.f;
    }
    float acc[MR][4];
#pragma unroll
    for (int r = 0; r < MR; r++) { acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.f; }
#pragma unroll
    for (int r = 0; r < MR; r++) {
      if ((deadm >> r) & 1) continue;               // (rows whose window has ended cost nothing)
#pragma unroll
      for (int i4 = 0; i4 < NR; i4 += 4) {
        const float4 xv = *reinterpret_cast<const float4*>(&xs[r][(2 * wave + hh) * NR + i4]);
        acc[r][0] += xv.x * w[i4].x; acc[r][1] += xv.x * w[i4].y; acc[r][2] += xv.x * w[i4].z; acc[r][3] += xv.x * w[i4].w;
        acc[r][0] += xv.y * w[i4 + 1].x; acc[r][1] += xv.y * w[i4 + 1].y; acc[r][2] += xv.y * w[i4 + 1].z; acc[r][3] += xv.y * w[i4 + 1].w;
        acc[r][0] += xv.z * w[i4 + 2].x; acc[r][1] += xv.z * w[i4 + 2].y; acc[r][2] += xv.z * w[i4 + 2].z; acc[r][3] += xv.z * w[i4 + 2].w;
        acc[r][0] += xv.w * w[i4 + 3].x; acc[r][1] += xv.w * w[i4 + 3].y; acc[r][2] += xv.w * w[i4 + 3].z; acc[r][3] += xv.w * w[i4 + 3].w;
      }
    }
#pragma unroll
    for (int r = 0; r < MR; r++)
#pragma unroll
      for (int c = 0; c < 4; c++) acc[r][c] = xor32_sum(acc[r][c]);        // the two row halves of the wave
    __syncthreads();                                // (the previous tile's readers of red / tilev are done)
    if (hh == 0) {
#pragma unroll
      for (int r = 0; r < MR; r++)
        *reinterpret_cast<float4*>(&red[wave][r][c4]) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int e = tid + NT * q, r = e / CT, c = e & (CT - 1);
      float v = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < 8; w8++) v += red[w8][r][c];
      v += mk[q];
      if (n0 + c >= a.V) v = -INFINITY;
      tilev[r][c] = v;
    }
    __syncthreads();
    if (wave < MR && !((deadm >> wave) & 1)) {
      const int r = wave;
      const float v0 = tilev[r][lane], v1 = tilev[r][lane + 64];
      float bvv; int bi;
      if (better(v0, lane, v1, lane + 64)) { bvv = v0; bi = lane; } else { bvv = v1; bi = lane + 64; }
      wave_argmax(bvv, bi);
      if (lane == 0) {
        float* ts = a.tstats + ((int64_t)r * a.n_tiles + tile) * 2;
        st_f<true>(ts, bvv);
        st_f<true>(ts + 1, __int_as_float(n0 + bi));
      }
    }
  };
  run_tile(w0, tile0);
  ps_stamp(ps, 4);
  if constexpr (TWO) {
    if (n_t > 1) run_tile(w1, tile0 + 1);
    ps_stamp(ps, 5);
    for (int t = 2; t < n_t; t++) { load_tile(w0, tile0 + t); run_tile(w0, tile0 + t); }
  } else {
    for (int t = 1; t < n_t; t++) { load_tile(w0, tile0 + t); run_tile(w0, tile0 + t); }
  }
  return true;
}

// ---- merge role: row r's argmax over the tiles, the chain's bookkeeping, the next step's embedding -------------------
// (what dec_topk_merge_kernel + chained_update do between two launches; transcribe.rs:235-241 for the end of a row)
__device__ __forceinline__ bool ps_merge_role(const PersistArgs& a, const int r, const int e, const PsStep& ps0,
                                              const unsigned* clog, const int* n_per_ctr) {
  __shared__ float redv[8];
  __shared__ int redi[8];
  const int tid = role_tid<true>(), lane = tid & 63, wave = tid >> 6;
  const PsStep& ps = ps0;
  {
    // every logits role of this step has arrived (8 sharded counters, polled by 8 lanes at once)
    __shared__ unsigned tgt[8];
    if (tid < 8) tgt[tid] = (unsigned)(e + 1) * (unsigned)n_per_ctr[tid];
    __syncthreads();
    if (!hx_wait_many(clog, tgt, 8, ps.ctl, ps.step, 1000, ps.lds_flag)) return false;
  }
  ps_stamp(ps, 1);
  const int len = ps.step + 1;                      // tokens in the row so far; the new one lands at index len
  const int fin_now = ld_i<true>(a.dead + r);
  const int tok_prev = ld_i<true>(a.gctl + GC_HDR + r);
  int gi;
  if (e < a.n_forced) {
    gi = a.forced[e];                               // prompt prefill: the next position holds the next prompt token
  } else {
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int t = tid; t < a.n_tiles; t += PS_NT) {
      const float* ts = a.tstats + ((int64_t)r * a.n_tiles + t) * 2;
      const float v = ld_f<true>(ts);
      const int id = __float_as_int(ld_f<true>(ts + 1));
      if (better(v, id, bv, bi)) { bv = v; bi = id; }
    }
    wave_argmax(bv, bi);
    if (lane == 0) { redv[wave] = bv; redi[wave] = bi; }
    __syncthreads();
    float gv = redv[0]; gi = redi[0];
#pragma unroll
    for (int j = 1; j < 8; j++)
      if (better(redv[j], redi[j], gv, gi)) { gv = redv[j]; gi = redi[j]; }
    // no finite candidate (NaN logits): never an embedding index -- the row ends, the host fails the call (decode.hip: top1_or_eot)
    if ((unsigned)gi >= 0x7fffffffu) { gi = a.eot; if (tid == 0) a.gctl[GC_BAD] = 1; }
  }
  const int finished = fin_now || (gi == a.eot && e >= a.n_forced);
  const int tok_next = fin_now ? tok_prev : gi;     // a finished row keeps its last token (its later argmax is stale)
  if (tid == 0 && !fin_now) {
    st_i_sc1(a.gctl + GC_HDR + r, gi);
    a.gtok[r * a.Lmax + len] = gi;
    a.gctl[GC_HDR + 2 * a.S + r] = len + 1;
    if (gi == a.eot && e >= a.n_forced) {           // transcribe.rs:235-241: the row has ended
      a.gctl[GC_HDR + a.S + r] = 1;
      st_i_sc1(a.dead + r, 1);
      const unsigned n = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(a.ctl + HX_NDONE), 1u, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT) + 1u;
      if ((int)n == ps.n_rows) st_i_sc1(a.ctl + HX_STOP, ps.step + 1);     // every window has ended: no further step
    }
  }
  if (tid == 0 && r == 0) a.gctl[GC_STEP] = ps.step + 1;
  if (len < a.Lmax) {
    // next step: position len holds tok_next -> x0[r] = E[tok_next] + pos[len] (mod.rs:141-146); position tables
    const Buf16 xb(a.x0);
    const float4* ev = reinterpret_cast<const float4*>(a.E + (int64_t)tok_next * a.d);
    const float4* pp = reinterpret_cast<const float4*>(a.pos + (int64_t)len * a.d);
    const unsigned tag_x = a.tag_base + 1u + 3u * (unsigned)((e + 1) * a.n_layer);   // what attn(0) of step e + 1 expects
    for (int c = tid; c < (a.d >> 2); c += PS_NT) {
      const float4 x = ev[c], y = pp[c];
      st_gran4(xb, (uint32_t)(r * a.d + 4 * c), tag_x, make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w));
    }
    // (the persistent roles address the self-attention cache arithmetically; the tables are kept for host-driven steps
    // that may follow the chain)
    const int nstep = ps.step + 1;
    int* tab_new = a.tabs + (size_t)(nstep & 1) * a.S * a.Lmax;
    for (int p = tid; p <= len; p += PS_NT) tab_new[r * a.Lmax + p] = p * a.S + r;
  }
  (void)finished;
  return true;
}

// ---- the grid ---------------------------------------------------------------------------------------------------------
template <int DPL, int MR, int NP>
__global__ __launch_bounds__(PS_NT) void dec_persist_kernel(PersistArgs a) {
  __shared__ int wait_flag;
  const int NL = a.n_layer, S = a.S, R = a.n_rows, H = a.n_head, NB = a.nb_mlp;
  // arrival counter c lives at ctl[HX_HDR + c HX_LINE]: one 128-byte line each
  auto cptr = [&](int c) { return reinterpret_cast<unsigned*>(a.ctl + HX_HDR + c * HX_LINE); };
  const int C_X = 0, C_ATTN = S, C_CROSS = S + NL * S, C_MLP = C_CROSS + NL, C_LOG = C_MLP + NL, C_GO = C_LOG + 8;
  int n_per_ctr[8];
#pragma unroll
  for (int k = 0; k < 8; k++) n_per_ctr[k] = (a.n_logits_roles + 7 - k) / 8;
  const int i_lo = a.role_off[blockIdx.x], i_hi = a.role_off[blockIdx.x + 1];
  unsigned long long dead_seen = 0;                 // bit k: role i_lo + k of this block has found its row dead (it stays dead)
  for (int e = 0; e < a.n_steps; e++) {
    for (int i = i_lo; i < i_hi; i++) {
      const PsRole role = a.roles[i];
      PsStep ps;
      ps.known_dead = i - i_lo < 64 && ((dead_seen >> (i - i_lo)) & 1ull) != 0;
      ps.ctl = a.ctl; ps.step = a.step0 + e; ps.n_rows = R; ps.dead = a.dead; ps.lds_flag = &wait_flag;
      unsigned long long* stp = a.stamps ? a.stamps + ((size_t)e * a.n_roles + i) * PS_STAMPS : nullptr;
      if (stp && threadIdx.x == 0) stp[0] = wall_clock64();
      ps.stamp = stp;
      int out;
      bool ok;
      // granule tags: a stage's planes and folded stream carry tag(e, layer, sublayer); a role consumes tag - 1
      const unsigned tag_l = a.tag_base + 2u + 3u * (unsigned)(e * NL + (role.kind <= PSR_MLP ? role.layer : NL - 1));
      if (role.kind == PSR_ATTN) {
        const AttnFusedArgs la = a.layers[role.layer].attn;
        // layer 0: the merge role's flag (its x row follows as granules); else pre-wake = the previous layer's
        // cross-attention blocks have finished (its MLP, the producer, is running)
        if (role.layer == 0) { ps.ctr_index = C_X + role.b; ps.target = (unsigned)e; }
        else { ps.ctr_index = C_CROSS + role.layer - 1; ps.target = (unsigned)(e + 1) * H * R; }
        ps.ctr = cptr(ps.ctr_index);
        ps.tag_out = tag_l; ps.tag_in = tag_l - 1u;
        ok = dec_attn_body<DPL, true>(la, role.a, role.b, ps);
        out = C_ATTN + role.layer * S + role.b;
      } else if (role.kind == PSR_CROSS) {
        const CrossFusedArgs la = a.layers[role.layer].cross;
        // pre-wake = the stage before the self-attention blocks: the previous layer's MLP (layer 0: the merge role)
        if (role.layer == 0) { ps.ctr_index = C_X + role.b; ps.target = (unsigned)e; }
        else { ps.ctr_index = C_MLP + role.layer - 1; ps.target = (unsigned)(e + 1) * NB; }
        ps.ctr = cptr(ps.ctr_index);
        ps.tag_out = tag_l + 1u; ps.tag_in = tag_l;
        ok = dec_cross_body<DPL, true, NP>(la, role.a, role.b, ps);
        out = C_CROSS + role.layer;
      } else if (role.kind == PSR_MLP) {
        const MlpFusedArgs la = a.layers[role.layer].mlp;
        // pre-wake = the self-attention blocks of every row have finished (the cross-attention blocks are running)
        ps.ctr_index = C_ATTN + role.layer * S; ps.target = (unsigned)(e + 1) * H; ps.n_ctr = R;
        ps.ctr = cptr(ps.ctr_index);
        ps.tag_out = tag_l + 2u; ps.tag_in = tag_l + 1u;
        ok = dec_mlp_body<MR, DPL, false, true>(la, role.a, ps);
        out = C_MLP + role.layer;
      } else if (role.kind == PSR_FINLN) {
        ps.ctr_index = C_CROSS + NL - 1; ps.target = (unsigned)(e + 1) * H * R;
        ps.ctr = cptr(ps.ctr_index);
        ps.tag_in = tag_l + 2u; ps.tag_out = tag_l + 2u;
        ok = ps_finln_role<DPL>(a, role.b, ps);
        out = C_GO + PS_NGO;                         // (nobody waits for it: the logits roles watch the granules)
      } else if (role.kind == PSR_LOGITS) {
        // a few hundred blocks: pre-woken by the last cross-attention block of the step through one of PS_NGO words; they
        // nap, then watch the final-LN roles' d granules per row (a sweep of all 4 d / 64 MLP planes by every logits block
        // was 22 MB per round: profiles/r03_c_ps_timeline_granules_v1.txt)
        ps.ctr_index = C_GO + (role.layer % PS_NGO); ps.target = (unsigned)(e + 1);
        ps.ctr = cptr(ps.ctr_index);
        ps.tag_in = tag_l + 2u;
        ok = ps_logits_role<MR, DPL>(a, role.a, role.b, ps, e < a.n_forced);
        out = C_LOG + (role.layer & 7);
      } else {
        ok = ps_merge_role(a, role.b, e, ps, cptr(C_LOG), n_per_ctr);
        out = C_X + role.b;
      }
      if (!ok) return;                               // the decode was stopped (or a wait gave up): leave
      if (ps.saw_dead && i - i_lo < 64) dead_seen |= 1ull << (i - i_lo);
      if (stp && threadIdx.x == 0) stp[6] = wall_clock64();
      if (role.kind == PSR_CROSS && role.layer == NL - 1)
        hx_arrive_broadcast(cptr(out), (unsigned)(e + 1) * H * R, cptr(C_GO), PS_NGO, (unsigned)(e + 1));
      else
        hx_arrive(cptr(out));
      if (stp && threadIdx.x == 0) stp[7] = wall_clock64();
    }
  }
}

// the first step's x rows (dec_prepare_kernel wrote plain floats) as granules with the tag attn(0) of step 0 expects
__global__ void ps_seed_kernel(const float* __restrict__ x, int n, void* gx, unsigned tag) {
  const Buf16 b(gx);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) st_gran(b, (uint32_t)i, tag, x[i]);
}

template <int DPL, int MR, int NP>
int max_blocks_per_cu() {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (dec_persist_kernel<DPL, MR, NP>), PS_NT, 0) != hipSuccess) return 0;
  return nb;
}

}  // namespace

int ps_ctl_ints(int S, int n_layer) { return HX_HDR + (S + n_layer * S + 2 * n_layer + 8 + PS_NGO + 1) * HX_LINE; }

// d = 512 with more than 4 rows would need > 160 KB of LDS (every role's LDS is resident at once); so would the two-pass
// cross-attention role (768 < C <= 1536 keys: the opt-in 30 s window) next to the 8-row MLP role
bool dec_persist_supported(int d, int n_rows, int max_keys) {
  if (n_rows < 1 || n_rows > 8) return false;
  if (max_keys > CROSS_FUSED_MAX_PASSES * CROSS_FUSED_MAX_C) return false;
  if (max_keys > CROSS_FUSED_MAX_C && n_rows > 4) return false;
  if (d == 128 || d == 384) return true;
  return d == 512 && n_rows <= 4;
}

int dec_persist_max_grid(int device, int d, int n_rows, int max_keys) {
  if (!dec_persist_supported(d, n_rows, max_keys)) return 0;
  int cus = 0, coop = 0;
  if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, device) != hipSuccess || !coop) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) return 0;
  int per = 0;
  const bool big = n_rows > 4;
  if (max_keys > CROSS_FUSED_MAX_C)
    per = d == 128 ? max_blocks_per_cu<2, 4, 2>() : d == 384 ? max_blocks_per_cu<6, 4, 2>() : max_blocks_per_cu<8, 4, 2>();
  else if (d == 128) per = big ? max_blocks_per_cu<2, 8, 1>() : max_blocks_per_cu<2, 4, 1>();
  else if (d == 384) per = big ? max_blocks_per_cu<6, 8, 1>() : max_blocks_per_cu<6, 4, 1>();
  else per = max_blocks_per_cu<8, 4, 1>();
  if (per <= 0) return 0;
  // (one block per CU is what the roles are sized for; a second resident block per CU would only share its fill path)
  return cus * 1;
}

void launch_ps_seed(hipStream_t st, const float* x, int n, void* gx, unsigned tag) {
  hipLaunchKernelGGL(ps_seed_kernel, dim3((n + 255) / 256), dim3(256), 0, st, x, n, gx, tag);
}

int launch_dec_persist(hipStream_t st, const PersistArgs& a, int grid) {
  const dim3 g(grid), b(PS_NT);
  const bool big = a.n_rows > 4;
  hipError_t e = hipErrorInvalidValue;
#define WB_PS(DPL_, MR_, NP_) e = WB_LAUNCH_COOP((dec_persist_kernel<DPL_, MR_, NP_>), g, b, 0, st, a)
  if (a.n_pass > 1) {
    if (big) return -1;
    if (a.d == 128) WB_PS(2, 4, 2); else if (a.d == 384) WB_PS(6, 4, 2); else if (a.d == 512) WB_PS(8, 4, 2);
  }
  else if (a.d == 128) { if (big) WB_PS(2, 8, 1); else WB_PS(2, 4, 1); }
  else if (a.d == 384) { if (big) WB_PS(6, 8, 1); else WB_PS(6, 4, 1); }
  else if (a.d == 512 && !big) WB_PS(8, 4, 1);
#undef WB_PS
  return e == hipSuccess ? 0 : -1;
}

}  // namespace wb
