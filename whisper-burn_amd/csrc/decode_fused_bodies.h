// Bodies of the fused small-batch decode sublayers (device templates), shared by
//   decode_fused.hip    one launch per sublayer (host-driven beam steps, the graph-replayed chain)
//   decode_persist.hip  the persistent flag-chained decode kernel: one co-resident grid runs every sublayer of every
//                       step; a role streams its weights BEFORE it waits for its producers' arrival counter
// PS = false: launch mode (the step state is read from a.st, plain loads / stores).  PS = true: persistent mode (`ps`
// carries the dependency counter and the step; data other blocks wrote in this launch is read with sc1 loads and
// published with sc1 stores -- handoff.h).  A body returns false only in persistent mode, when the decode was stopped
// (the caller leaves the kernel); the caller arrives at the role's output counter after a `true` return.
#pragma once
#include <hip/hip_runtime.h>

#include "decode.h"
#include "handoff.h"
#include "wave_ops.h"

namespace wb {
namespace fused {

// what a role needs to know about the chain step it runs in (persistent mode only)
struct PsStep {
  const unsigned* ctr = nullptr; unsigned target = 0; int ctr_index = 0;   // the dependency: *ctr >= target
  const int* ctl = nullptr;      // control words of the launch (HX_STOP, HX_ERR, ...)
  int step = 0;                  // decode step = position of the new token; every row has length step + 1
  int n_rows = 0;                // rows of the chain (= windows: one beam per window)
  const int* dead = nullptr;     // [S]: rows whose window has ended (written by the merge role, sc1)
  int* lds_flag = nullptr;       // one LDS word for the wait's broadcast
  unsigned long long* stamp = nullptr;   // optional timeline slots of this role invocation (PS_STAMPS of them)
  int n_ctr = 1;                 // the wait covers n_ctr consecutive counters (HX_LINE apart), all with `target`
  // granule tags (handoff.h): what the role's producers stamp their planes with / what the role stamps its own output with
  unsigned tag_in = 0, tag_out = 0;
  // a row's window has ended: in = the block has seen this role's row dead in an earlier step (it stays dead: the role
  // requests nothing before its wait -- the weight / cached-K prefetch of dead rows was most of the launch's traffic above
  // the necessary bytes, profiles/r03_d_pmc_traffic_tiny_en_30s.json); out = the role found its row dead this step
  bool known_dead = false;
  mutable bool saw_dead = false;
};
constexpr int PS_STAMPS = 8;     // 0 role start, 1 wait passed, 2-5 phases inside the role, 6 done, 7 arrived
__device__ __forceinline__ void ps_stamp(const PsStep& ps, int k) {
#ifdef WB_EXP_FINE
  if (k == 4 || k == 5) return;                    // (developer build: slots 4 and 5 resolve the first phase instead)
#endif
  if (ps.stamp && threadIdx.x == 0) ps.stamp[k] = wall_clock64();
}
__device__ __forceinline__ void ps_stamp_fine(const PsStep& ps, int k) {
#ifdef WB_EXP_FINE
  if (ps.stamp && threadIdx.x == 0) ps.stamp[k] = wall_clock64();
#endif
}
// wait for the role's producers (all threads of the block); false: the decode was stopped -- leave the kernel
__device__ __forceinline__ bool ps_wait(const PsStep& ps) {
  const bool ok = ps.n_ctr > 1 ? hx_wait_same(ps.ctr, ps.target, ps.n_ctr, ps.ctl, ps.step, ps.ctr_index, ps.lds_flag)
                               : hx_wait(ps.ctr, ps.target, ps.ctl, ps.step, ps.ctr_index, ps.lds_flag);
  ps_stamp(ps, 1);
  return ok;
}
// A granule sweep found stale tags: give way, and (rarely) look at the stop / error words.  true = give up -- the caller
// clears *ps.lds_flag, which every thread of the block reads behind the next barrier (ps_sweeps_ok).
__device__ __forceinline__ bool ps_sweep_retry(unsigned& sweeps, const PsStep& ps) {
  if ((sweeps & 31u) == 31u &&
      (__hip_atomic_load(ps.ctl + HX_STOP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= ps.step ||
       __hip_atomic_load(ps.ctl + HX_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) return true;
  if (++sweeps > HX_SWEEP_LIMIT) {
    __hip_atomic_store(const_cast<int*>(ps.ctl) + HX_ERR, 5000 + ps.ctr_index, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
  }
  __builtin_amdgcn_s_sleep(1);
  return false;
}
__device__ __forceinline__ bool ps_sweeps_ok(const PsStep& ps) { return *ps.lds_flag != 0; }   // call BEHIND a barrier
// The pre-wake fires when the producers START; their planes land one stage later.  Napping through most of that stage
// keeps a block's re-reads (tens of KB per sweep) out of the memory system while nothing can have arrived yet.  Purely a
// matter of load: correctness never depends on it.  (s_sleep 64 ~ 1.7 us)
template <int UNITS>
__device__ __forceinline__ void ps_nap() {
#if !defined(WB_PS_DRY) || defined(WB_PS_DRY_NAPS)    // (dry build: the producers' bodies are empty -- nothing to sleep through)
#ifndef WB_PS_NAP_ADJ
#define WB_PS_NAP_ADJ 0                               // developer A/B: nap units added to every consumer's nap
#endif
#pragma unroll
  for (int i = 0; i < UNITS + (WB_PS_NAP_ADJ); i++) __builtin_amdgcn_s_sleep(64);
#endif
}

// developer probe: tools/decode_probe.cpp builds this file with -DWB_STAMPS and prints the phase timeline of block 0
#ifdef WB_STAMPS
#define WB_STAMP_DECL __shared__ unsigned long long stamp_buf[16]
#define WB_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) stamp_buf[i] = wall_clock64(); } while (0)
#define WB_STAMP_FLUSH(a, n) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && (a).stamps) for (int _i = 0; _i < (n); _i++) (a).stamps[_i] = stamp_buf[_i]; } while (0)
#else
#define WB_STAMP_DECL
#define WB_STAMP(i) do {} while (0)
#define WB_STAMP_FLUSH(a, n) do {} while (0)
#endif

constexpr int FD_MAX = 512;      // largest n_state of the fused path (test models 128, tiny.en 384, base.en 512)
constexpr int FA_MAXPOS = 448;   // n_text_ctx

__device__ __forceinline__ float gelu_erf_f(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// One wave folds row `row`: v = x_in + (pbias + sum_s pend[s]) (s ascending, mod.rs:346-348), optionally writes
// the folded stream, then LayerNorm (Burn nn::LayerNorm: biased variance, two passes) into out[0..d).
// Every load is unconditional (columns past d alias column `lane`; planes past KSp alias the last plane).
template <int DPL>
__device__ __forceinline__ void fold_ln_row(const float* __restrict__ x_in, const float* __restrict__ pend, int KSp,
                                            int64_t plane, const float* __restrict__ pbias, float* __restrict__ x_out,
                                            const float* __restrict__ g, const float* __restrict__ b, float eps,
                                            int eps_inside, int d, int row, int lane, float* __restrict__ out) {
  int co[DPL];
#pragma unroll
  for (int i = 0; i < DPL; i++) co[i] = lane + (64 * i < d ? 64 * i : 0);
  float gv[DPL], bv[DPL], v[DPL];
  const float* xr = x_in + (int64_t)row * d;
#pragma unroll
  for (int i = 0; i < DPL; i++) { gv[i] = g[co[i]]; bv[i] = b[co[i]]; v[i] = xr[co[i]]; }
  if (KSp > 0) {
    float acc[DPL];
#pragma unroll
    for (int i = 0; i < DPL; i++) acc[i] = pbias[co[i]];
    const float* pp = pend + (int64_t)row * d;
    constexpr int CH = DPL <= 6 ? 8 : 4;
    for (int sp = 0; sp < KSp; sp += CH) {
      float t[CH][DPL];
#pragma unroll
      for (int j = 0; j < CH; j++) {
        const float* pj = pp + (int64_t)min(sp + j, KSp - 1) * plane;
#pragma unroll
        for (int i = 0; i < DPL; i++) t[j][i] = pj[co[i]];
      }
#pragma unroll
      for (int j = 0; j < CH; j++) {
        const bool live = sp + j < KSp;
#pragma unroll
        for (int i = 0; i < DPL; i++) acc[i] += live ? t[j][i] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < DPL; i++) v[i] = v[i] + acc[i];
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < DPL; i++) {
    if (64 * i < d) { s += v[i]; if (x_out) x_out[(int64_t)row * d + co[i]] = v[i]; }
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < DPL; i++) {
    if (64 * i < d) { const float t = v[i] - mean; q += t * t; }
  }
  const float var = wave_sum(q) / (float)d;
  const float denom = eps_inside ? sqrtf(var + eps) : (sqrtf(var) + eps);
#pragma unroll
  for (int i = 0; i < DPL; i++) {
    const int c = lane + 64 * i;
    if (c < d) out[c] = (v[i] - mean) / denom * gv[i] + bv[i];
  }
}

// Cooperative fold of the residual stream: xs[r][c] = x_in[r][c] + pbias[c] + sum_s pend[s][r][c] for every
// r < MR, c < d, all loads of a thread issued together (one memory round trip instead of one per plane chunk).
// Rows >= n_rows read valid memory (the buffers carry MR rows of slack) and are never used.
template <int NT, int MR, int EPT, int PCH>
__device__ __forceinline__ void fold_rows(const float* __restrict__ x_in, const float* __restrict__ pend, int KSp,
                                          int64_t plane, const float* __restrict__ pbias, int d, int tid,
                                          float (&v)[EPT]) {
  int off[EPT], col[EPT];
#pragma unroll
  for (int i = 0; i < EPT; i++) {
    int e = tid + NT * i;
    if (e >= MR * d) e = tid;                      // past the tile: alias an in-range element, result unused
    off[i] = e; col[i] = e % d;
  }
#pragma unroll
  for (int i = 0; i < EPT; i++) v[i] = x_in[off[i]];
  if (KSp > 0) {
    float acc[EPT];
#pragma unroll
    for (int i = 0; i < EPT; i++) acc[i] = pbias[col[i]];
    for (int sp = 0; sp < KSp; sp += PCH) {
      float t[PCH][EPT];
#pragma unroll
      for (int j = 0; j < PCH; j++) {
        const float* pj = pend + (int64_t)min(sp + j, KSp - 1) * plane;
#pragma unroll
        for (int i = 0; i < EPT; i++) t[j][i] = pj[off[i]];
      }
#pragma unroll
      for (int j = 0; j < PCH; j++) {
        const bool live = sp + j < KSp;
#pragma unroll
        for (int i = 0; i < EPT; i++) acc[i] += live ? t[j][i] : 0.f;     // s ascending: fixed order
      }
    }
#pragma unroll
    for (int i = 0; i < EPT; i++) v[i] = v[i] + acc[i];                   // x + (bias + partials)  (mod.rs:346-348)
  }
}

// LayerNorm of row r held in LDS (in place), one wave: Burn nn::LayerNorm, biased variance, two passes.
template <int DPL>
__device__ __forceinline__ void ln_row_lds(float* __restrict__ row, int d, int lane, const float (&gv)[DPL],
                                           const float (&bv)[DPL], float eps, int eps_inside) {
  float v[DPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < DPL; i++) { v[i] = row[lane + 64 * i]; s += v[i]; }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < DPL; i++) { const float t = v[i] - mean; q += t * t; }
  const float var = wave_sum(q) / (float)d;
  const float denom = eps_inside ? sqrtf(var + eps) : (sqrtf(var) + eps);
#pragma unroll
  for (int i = 0; i < DPL; i++) row[lane + 64 * i] = (v[i] - mean) / denom * gv[i] + bv[i];
}

// LayerNorm statistics of a row held in LDS, computed by EVERY wave for itself (same loads, same reductions as ln_row_lds:
// identical bits in every wave), so that a wave can normalise the few elements its own matrix rows need and go on without
// a second block barrier -- wave 0 normalising the whole row for everybody cost a barrier-delimited phase (~0.5 us of a
// 10 us role in the persistent kernel, profiles/r03_g_ps_fine.txt).  Returns mean and 1-divisor through `mean`, `denom`.
template <int DPL>
__device__ __forceinline__ void ln_stats_lds(const float* __restrict__ row, int d, int lane, float eps, int eps_inside,
                                             float& mean, float& denom) {
  float v[DPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < DPL; i++) { v[i] = row[lane + 64 * i]; s += v[i]; }
  mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < DPL; i++) { const float t = v[i] - mean; q += t * t; }
  const float var = wave_sum(q) / (float)d;
  denom = eps_inside ? sqrtf(var + eps) : (sqrtf(var) + eps);
}
// value of lane `src` (wave-uniform or per-lane) of a register: the LDS crossbar without LDS memory (ds_bpermute_b32)
__device__ __forceinline__ float lane_get(float v, int src) { return __shfl(v, src, 64); }

// ---------------------------------------------------------------------------------------------------------
// MLP.  block = 512 threads, grid = 4 d / 64.  Every global load of the block (fold operands, the W1 slice, the
// W2 slice) is requested before the first use of any of them: one memory round trip on the critical path.
// Phase 1: 16 lanes x float4 cover the 64 slice columns of one W1 row; thread (rg = tid / 16, c4) owns rows
// rg, rg + 32, ... (d / 32 of them).  Phase 2: thread (cf = tid % (d / 4), jg = tid / (d / 4)) owns a float4 of
// output columns and 64 / G rows of the W2 slice.
template <int MR, int DPL, bool REC, bool PS>
__device__ __forceinline__ bool dec_mlp_body(const MlpFusedArgs& a, const int jblk, const PsStep& ps) {
  constexpr int HS = 64, NT = 512;
  constexpr int d = 64 * DPL;
  constexpr int CF = d / 4;
  constexpr int G = CF <= 32 ? 8 : 4;              // phase-2 row groups (d = 128: 8 x 8 rows, 384 / 512: 4 x 16 rows)
  constexpr int RPG = HS / G;
  constexpr int EPT = (MR * d + NT - 1) / NT;
  constexpr int PCH = EPT <= 3 ? 8 : EPT <= 6 ? 6 : 3;
  __shared__ __attribute__((aligned(16))) float hs[MR][d];              // x + pending, then LN of it
  __shared__ __attribute__((aligned(16))) float red[8][MR][HS];         // per-wave partial hidden sums
  __shared__ __attribute__((aligned(16))) float hid[MR][HS];            // GELU(hidden slice)
  __shared__ __attribute__((aligned(16))) float obuf[(G - 1) * MR * d]; // phase-2 partials of row groups >= 1
  __shared__ float mlb[REC ? MR : 1][48][2];                            // record mode: (m, l) of every (head, chunk)
  __shared__ float coef[REC ? MR : 1][48];                              // ... and its flash-combine weight
  WB_STAMP_DECL;
  WB_STAMP(0);
  const int tid = role_tid<PS>(), lane = tid & 63, wave = tid >> 6;
  const int j0 = jblk * HS;
  const int rg = tid >> 4, c4 = (tid & 15) * 4;
  // Requested up front, in the order of use (loads return in order): fold operands, LayerNorm parameters, bias, the W1
  // slice, the W2 slice.  No global STORE before the last phase (a pending store makes every __syncthreads a vmcnt(0)).
  float xv_fold[EPT];
  constexpr int NW1 = d / 32;
  float4 w1[NW1];
  const int cf = tid % CF, jg = tid / CF;
  const bool p2 = jg < G;
  const int jb = (p2 ? jg : 0) * RPG;
  float4 w2[RPG];
  // LayerNorm parameters of the NW1 * 4 columns this wave's W1 rows need: lane l < 4 NW1 owns column 4 wave + (l & 3) +
  // 32 (l >> 2) = row rg + 32 i of thread group (rg = 4 wave + (l & 3), i = l >> 2)
  constexpr int NOWN = 4 * NW1;
  const int own_col = 4 * wave + (lane & 3) + 32 * ((lane < NOWN ? lane : 0) >> 2);
  float g_own = 0.f, b_own = 0.f;
  float b1v;
  int deadm = 0;                                   // (persistent mode) bit r: row r takes no part (its window has ended)
  {
    int off[EPT], col[EPT];
#pragma unroll
    for (int i = 0; i < EPT; i++) {
      int e = tid + NT * i;
      if (e >= MR * d) e = tid;
      off[i] = e; col[i] = e % d;
    }
    float acc0[EPT];
    constexpr bool rec = REC;                        // pend = cross-attention chunk records {m, l, P[d]}
    const int64_t plane = rec ? (int64_t)a.S * (d + 2) : (int64_t)a.S * d;
    int poff[EPT];                                   // element offset inside a plane / record plane
#pragma unroll
    for (int i = 0; i < EPT; i++) poff[i] = rec ? (off[i] / d) * (d + 2) + 2 + col[i] : off[i];
    constexpr int PCHR = EPT <= 3 ? 18 : EPT <= 4 ? 16 : EPT <= 6 ? 12 : 8;
    constexpr int PC = REC ? PCHR : PCH;
    float t[PC][EPT];
    const int npl = a.KSp;
    const int ch = rec ? PCHR : PCH;                 // planes per round
    auto load_weights = [&]() {                      // LayerNorm parameters, bias, the W1 slice, the W2 slice
      g_own = gld(a.ln_g + own_col); b_own = gld(a.ln_b + own_col);
      b1v = gld(a.b1 + j0 + (tid & 63));
      {
        const float* wp = a.W1 + (int64_t)rg * a.ld1 + j0 + c4;
#pragma unroll
        for (int i = 0; i < NW1; i++) w1[i] = ld_w4(wp + (int64_t)(32 * i) * a.ld1);
      }
      {
        const float* wp = a.W2 + (int64_t)(j0 + jb) * d + cf * 4;
#pragma unroll
        for (int i = 0; i < RPG; i++) w2[i] = ld_w4(wp + (int64_t)i * d);
      }
    };
    if constexpr (PS) {
      // persistent mode: nothing the block streams depends on its predecessor -- the weights are in flight (or landed)
      // while the block waits (pre-wake: the self-attention blocks of every row have finished); the cross-attention
      // planes arrive as tagged granules, re-read until every tag is the producers'
#if !defined(WB_PS_DRY)                             // (developer "dry" build, tools/build_exp.sh dry: every wait and hand-off of
      load_weights();                               //  a token with no weight / cache loads and no math -- the floor of this partition)
#endif
      if (!ps_wait(ps)) return false;
#pragma unroll
      for (int r = 0; r < MR; r++)
        if (r >= ps.n_rows || ld_i<true>(ps.dead + min(r, ps.n_rows - 1)) != 0) deadm |= 1 << r;
      const Buf16 xgb(a.g_x_in), pgb(a.g_pend);
      const int iplane = a.S * d;
      ps_nap<3>();                                  // (the cross-attention blocks have only just started)
      bool rdead[EPT];
#pragma unroll
      for (int i = 0; i < EPT; i++) { rdead[i] = ((deadm >> (off[i] / d)) & 1) != 0; acc0[i] = gld(a.pbias + col[i]); xv_fold[i] = 0.f; }
      constexpr int GP = EPT <= 3 ? 8 : 4;           // planes per sweep
      unsigned sweeps = 0;
      bool bad = false;
      for (int sp = 0; sp < npl && !bad; sp += GP) {
        Gran gx[EPT], g[GP][EPT];
#pragma unroll
        for (int i = 0; i < EPT; i++) {
          gx[i] = Gran{0u, 0.f};
#pragma unroll
          for (int j = 0; j < GP; j++) g[j][i] = Gran{0u, 0.f};
        }
        for (;;) {
          if (sp == 0) {
#pragma unroll
            for (int i = 0; i < EPT; i++)
              if (!rdead[i] && gx[i].tag != ps.tag_in) gx[i] = ld_gran(xgb, (uint32_t)off[i], 0u);
          }
#pragma unroll
          for (int j = 0; j < GP; j++)
#pragma unroll
            for (int i = 0; i < EPT; i++)
              if (!rdead[i] && sp + j < npl && g[j][i].tag != ps.tag_in)
                g[j][i] = ld_gran(pgb, (uint32_t)off[i], (uint32_t)((sp + j) * iplane));
          bool ok = true;
#pragma unroll
          for (int i = 0; i < EPT; i++) {
            if (sp == 0) ok &= rdead[i] || gx[i].tag == ps.tag_in;
#pragma unroll
            for (int j = 0; j < GP; j++) ok &= rdead[i] || sp + j >= npl || g[j][i].tag == ps.tag_in;
          }
          if (ok) {
#pragma unroll
            for (int i = 0; i < EPT; i++) {
              if (sp == 0) xv_fold[i] = gx[i].v;
#pragma unroll
              for (int j = 0; j < GP; j++) acc0[i] += (sp + j < npl) ? g[j][i].v : 0.f;      // plane order fixed
            }
            break;
          }
          if (ps_sweep_retry(sweeps, ps)) { *ps.lds_flag = 0; bad = true; break; }
        }
      }
#pragma unroll
      for (int i = 0; i < EPT; i++) {
        xv_fold[i] = rdead[i] ? 0.f : xv_fold[i] + acc0[i];                              // x + (bias + partials)  (mod.rs:346-348)
        const int e = tid + NT * i;
        if (e < MR * d) (&hs[0][0])[e] = xv_fold[i];
      }
    } else {
#pragma unroll
    for (int i = 0; i < EPT; i++) { xv_fold[i] = a.x_in[off[i]]; acc0[i] = npl > 0 ? gld(a.pbias + col[i]) : 0.f; }
    float mlv0 = -1.0e30f, mlv1 = 0.f;
    if (rec && tid < MR * npl) {                     // (m, l) of record (row tid / npl, plane tid % npl)
      const float* rp = a.pend + (int64_t)(tid % npl) * plane + (int64_t)(tid / npl) * (d + 2);
      mlv0 = rp[0]; mlv1 = rp[1];
    }
    if (npl > 0) {
#pragma unroll
      for (int j = 0; j < PC; j++)
        if (j < ch) {
#pragma unroll
          for (int i = 0; i < EPT; i++) t[j][i] = a.pend[(int64_t)min(j, npl - 1) * plane + poff[i]];
        }
    }
    load_weights();
    if constexpr (REC) {
      // flash-combine weights: coef[r][h][c] = exp(m_hc - M_h) / sum_c' exp(m_hc' - M_h) l_hc'  (mod.rs:529 softmax,
      // split over key chunks by dec_cross_attn_kernel)
      if (tid < MR * npl) { mlb[tid / npl][tid % npl][0] = mlv0; mlb[tid / npl][tid % npl][1] = mlv1; }
      __syncthreads();
      if (tid < MR * a.n_head) {
        const int r = tid / a.n_head, hh = tid % a.n_head;
        float M = -1.0e30f;
        for (int c = 0; c < a.n_chunks; c++) M = fmaxf(M, mlb[r][hh * a.n_chunks + c][0]);
        float den = 0.f;
        for (int c = 0; c < a.n_chunks; c++) den += expf(mlb[r][hh * a.n_chunks + c][0] - M) * mlb[r][hh * a.n_chunks + c][1];
        for (int c = 0; c < a.n_chunks; c++)
          coef[r][hh * a.n_chunks + c] = den > 0.f ? expf(mlb[r][hh * a.n_chunks + c][0] - M) / den : 0.f;
      }
      __syncthreads();
    }
    if (npl > 0) {
      int rowi[EPT];
#pragma unroll
      for (int i = 0; i < EPT; i++) rowi[i] = off[i] / d;
      for (int sp = 0; sp < npl; sp += ch) {
        if (sp > 0) {
#pragma unroll
          for (int j = 0; j < PC; j++)
            if (j < ch) {
#pragma unroll
              for (int i = 0; i < EPT; i++) t[j][i] = a.pend[(int64_t)min(sp + j, npl - 1) * plane + poff[i]];
            }
        }
#pragma unroll
        for (int j = 0; j < PC; j++)
          if (j < ch) {
#pragma unroll
            for (int i = 0; i < EPT; i++) {
              const bool live = sp + j < npl;
              float wgt = 1.f;
              if constexpr (REC) wgt = coef[rowi[i]][min(sp + j, npl - 1)];
              acc0[i] += live ? wgt * t[j][i] : 0.f;                                        // plane order fixed
            }
          }
      }
#pragma unroll
      for (int i = 0; i < EPT; i++) xv_fold[i] += acc0[i];                               // x + (bias + partials)  (mod.rs:346-348)
    }
#pragma unroll
    for (int i = 0; i < EPT; i++) {
      const int e = tid + NT * i;
      if (e < MR * d) (&hs[0][0])[e] = xv_fold[i];
    }
    }
  }
  int n_rows;
  if constexpr (PS) n_rows = ps.n_rows; else n_rows = min(MR, a.st[ST_N] - a.row0);   // (row group: rows [row0, row0 + MR))
  if (n_rows <= 0) return true;                    // chained decode, every window finished / an empty row group (block-uniform)
  WB_STAMP(1);
  __syncthreads();
  if constexpr (PS) { if (!ps_sweeps_ok(ps)) return false; }
#if defined(WB_PS_DRY)
  if constexpr (PS) {                                // dry build: publish what the role publishes (zeros), nothing else
    if (jg == 0) {
      const Buf16 pgo(a.g_P);
#pragma unroll
      for (int r = 0; r < MR; r++)
        if (r < n_rows && !((deadm >> r) & 1)) st_gran4(pgo, (uint32_t)((jblk * a.S + r) * d + cf * 4), ps.tag_out, make_float4(0.f, 0.f, 0.f, 0.f));
    }
    if (jblk == 0) {
      const Buf16 xgo(a.g_x_out);
#pragma unroll
      for (int i = 0; i < EPT; i++) {
        const int e = tid + NT * i;
        if (e < n_rows * d && !((deadm >> (e / d)) & 1)) st_gran(xgo, (uint32_t)e, ps.tag_out, xv_fold[i]);
      }
    }
    return true;
  }
#endif
  // LayerNorm: every wave takes the statistics of every (live) row itself and normalises the columns its own W1 rows
  // multiply -- no second barrier; the normalised values stay in registers (lane l: column own_col) and reach the
  // thread groups through the LDS crossbar (lane 4 i + q holds row rg + 32 i of group q = lane / 16)
  float xn_own[MR];
#pragma unroll
  for (int r = 0; r < MR; r++) {
    xn_own[r] = 0.f;
    bool skip = r >= n_rows;
    if constexpr (PS) skip = skip || ((deadm >> r) & 1);
    if (skip) continue;                              // (block-uniform)
    float mean, denom;
    ln_stats_lds<DPL>(hs[r], d, lane, a.ln_eps, a.ln_inside, mean, denom);
    xn_own[r] = (hs[r][own_col] - mean) / denom * g_own + b_own;
  }
  WB_STAMP(2);
  float acc[MR][4];
#pragma unroll
  for (int r = 0; r < MR; r++) { acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.f; }
  {
    const int q4 = lane >> 4;
    // (rows that take no part cost nothing: block-uniform branches around their FMAs)
#pragma unroll
    for (int r = 0; r < MR; r++) {
      bool skip = r >= n_rows;
      if constexpr (PS) skip = skip || ((deadm >> r) & 1);
      if (skip) continue;
#pragma unroll
      for (int i = 0; i < NW1; i++) {
        const float xv = lane_get(xn_own[r], 4 * i + q4);
        acc[r][0] += xv * w1[i].x; acc[r][1] += xv * w1[i].y; acc[r][2] += xv * w1[i].z; acc[r][3] += xv * w1[i].w;
      }
    }
  }
  // lanes l, l ^ 16, l ^ 32, l ^ 48 hold the same columns for different rows: fold them, then the eight waves
#pragma unroll
  for (int r = 0; r < MR; r++)
#pragma unroll
    for (int c = 0; c < 4; c++) { acc[r][c] = xor16_sum(acc[r][c]); acc[r][c] = xor32_sum(acc[r][c]); }
  if (lane < 16) {
#pragma unroll
    for (int r = 0; r < MR; r++)
      *reinterpret_cast<float4*>(&red[wave][r][c4]) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
  }
  __syncthreads();
  if (tid < MR * HS) {
    const int r = tid >> 6, c = tid & 63;
    float v = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < 8; w8++) v += red[w8][r][c];                  // wave order fixed
    hid[r][c] = gelu_erf_f(v + b1v);                                    // mod.rs:377-378
  }
  __syncthreads();
  WB_STAMP(3);
  float o[MR][4];
#pragma unroll
  for (int r = 0; r < MR; r++) { o[r][0] = o[r][1] = o[r][2] = o[r][3] = 0.f; }
  if constexpr (PS) {
#pragma unroll
    for (int r = 0; r < MR; r++) {
      if ((deadm >> r) & 1) continue;
#pragma unroll
      for (int i = 0; i < RPG; i++) {
        const float hv = hid[r][jb + i];
        o[r][0] += hv * w2[i].x; o[r][1] += hv * w2[i].y; o[r][2] += hv * w2[i].z; o[r][3] += hv * w2[i].w;
      }
    }
  } else {
#pragma unroll
  for (int i = 0; i < RPG; i++) {
#pragma unroll
    for (int r = 0; r < MR; r++) {
      const float hv = hid[r][jb + i];
      o[r][0] += hv * w2[i].x; o[r][1] += hv * w2[i].y; o[r][2] += hv * w2[i].z; o[r][3] += hv * w2[i].w;
    }
  }
  }
  if (p2 && jg > 0) {
#pragma unroll
    for (int r = 0; r < MR; r++)
      *reinterpret_cast<float4*>(&obuf[((jg - 1) * MR + r) * d + cf * 4]) = make_float4(o[r][0], o[r][1], o[r][2], o[r][3]);
  }
  __syncthreads();
  if (jg == 0) {
#pragma unroll
    for (int g2 = 1; g2 < G; g2++) {               // group order fixed: deterministic sums
#pragma unroll
      for (int r = 0; r < MR; r++) {
        const float4 t = *reinterpret_cast<const float4*>(&obuf[((g2 - 1) * MR + r) * d + cf * 4]);
        o[r][0] += t.x; o[r][1] += t.y; o[r][2] += t.z; o[r][3] += t.w;
      }
    }
    if constexpr (PS) {
      const Buf16 pgo(a.g_P);
#pragma unroll
      for (int r = 0; r < MR; r++)
        if (r < n_rows && !((deadm >> r) & 1))
          st_gran4(pgo, (uint32_t)((jblk * a.S + r) * d + cf * 4), ps.tag_out, make_float4(o[r][0], o[r][1], o[r][2], o[r][3]));
    } else {
#pragma unroll
      for (int r = 0; r < MR; r++)
        if (r < n_rows)
          *reinterpret_cast<float4*>(&a.P[((int64_t)jblk * a.S + r) * d + cf * 4]) = make_float4(o[r][0], o[r][1], o[r][2], o[r][3]);
    }
  }
  if (jblk == 0) {                                 // the folded residual stream, off the critical path
    if constexpr (PS) {
      const Buf16 xgo(a.g_x_out);
#pragma unroll
      for (int i = 0; i < EPT; i++) {
        const int e = tid + NT * i;
        if (e < n_rows * d && !((deadm >> (e / d)) & 1)) st_gran(xgo, (uint32_t)e, ps.tag_out, xv_fold[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < EPT; i++) {
        const int e = tid + NT * i;
        if (e < n_rows * d) a.x_out[e] = xv_fold[i];
      }
    }
  }
  WB_STAMP(4);
  WB_STAMP_FLUSH(a, 5);
  return true;
}

// ---------------------------------------------------------------------------------------------------------
// Self-attention block.  block = 512 threads (8 waves), grid = (n_head, rows): block (h, r) owns head h of beam r.
// (One block per head for ALL beams keeps 4-8 accumulator sets per thread next to two weight rounds and spills;
// per-beam blocks re-read the head's weight slices through L2 -- HBM still sees them once.)
// QKV: wave w owns K-rows [w d / 8, (w + 1) d / 8); lane (seg = lane / 16 in {q, k, v, idle}, c4) reads a float4 of
// the head's 64 columns of that segment; rounds of 16 rows, two rounds in flight.  Requested up front: the fold
// operands + two weight rounds, then (as soon as the position table is known) the cached K / V rows of the first 128
// positions in the coalesced row layout -- they land under the QKV FMAs; when the weight registers free up: the
// out-projection slice.  The attention phases and the out-projection do not wait on memory.
template <int DPL, bool PS>
__device__ __forceinline__ bool dec_attn_body(const AttnFusedArgs& a, const int h, const int r, const PsStep& ps) {
  constexpr int NT = 512;
  constexpr int d = 64 * DPL;
  constexpr int KW = d / 8;                        // K rows per wave (16, 48, 64)
  constexpr int RK = DPL >= 6 ? 8 : 16;            // K rows per weight round (sized so that two rounds + the cached K/V tile stay in registers)
  constexpr int NIT = KW / RK;
  constexpr int CF = d / 4;
  constexpr int G = CF <= 32 ? 8 : 4;              // out-projection row groups (d = 128: 8 x 8 rows, 384 / 512: 4 x 16)
  constexpr int RPG = 64 / G;
  constexpr int RED = (8 * 192 > (G - 1) * d) ? 8 * 192 : (G - 1) * d;
  constexpr int PT = 128, PSL = PT * 16 / NT;      // cached positions per tile, float4 slots per thread per tile (4)
  __shared__ __attribute__((aligned(16))) float hs[d];
  __shared__ __attribute__((aligned(16))) float red[RED];              // QKV partials, later the out-projection partials
  __shared__ __attribute__((aligned(16))) float qkv[192];              // q * s, k * s, v of the new token (head h)
  __shared__ int tbs[FA_MAXPOS];
  __shared__ float sc[FA_MAXPOS];
  __shared__ __attribute__((aligned(16))) float part[32][64];
  __shared__ __attribute__((aligned(16))) float att[64];
  WB_STAMP_DECL;
  WB_STAMP(0);
  const int tid = role_tid<PS>(), lane = tid & 63, wave = tid >> 6;
  // grid = (8, rows): workgroups go round-robin over the 8 XCDs by linear id, so x = head puts every beam's block of
  // one head on the SAME XCD -- the head's weight slices cross the fabric once and are shared through that L2
  if (h >= a.n_head) return true;
  // the row's state words first: the position table, and through it the cached K / V addresses, hang on them
  // (persistent mode: one beam per window, every row has the same length, cache row of position p = p S + r)
  int n_rows, len, step_par, dead;
  if constexpr (PS) { n_rows = ps.n_rows; len = ps.step + 1; step_par = 0; dead = 0; }
  else {
    n_rows = a.st[ST_N]; len = a.st[a.lay.len + r]; step_par = a.st[ST_STEP] & 1;
    dead = a.st[a.lay.dead + r];                   // chained greedy decode: this row's window has already ended
  }
  // ---- then, in this order (loads return in order: the fold must not queue behind the weights): the fold
  // operands of row r, LayerNorm parameters (wave 0 normalises), bias, then two QKV weight rounds.
  // No global STORE happens before the last phase: a pending store turns every __syncthreads into vmcnt(0).
  const int seg = lane >> 4, c4 = (lane & 15) * 4;
  const bool seg_ok = seg < 3;
  const float* wq = a.Wqkv + (int64_t)(wave * KW) * a.ldqkv + (seg_ok ? seg : 0) * d + h * 64 + c4;
  float4 wr[2][RK];
  auto load_round = [&](float4 (&w)[RK], int it) {
#pragma unroll
    for (int j = 0; j < RK; j++) w[j] = ld_w4(wq + (int64_t)(RK * it + j) * a.ldqkv);
  };
  float xfold;                                     // this thread's element of x + pending (kept for the final x_out store)
  // LayerNorm parameters of the KW rows of x this wave's QKV rounds multiply: lane l < KW owns row wave KW + l
  const int own_col = wave * KW + (lane < KW ? lane : 0);
  float g_own = 0.f, b_own = 0.f;
  float qbias = 0.f;
  // ---- the cached K / V rows of the first 128 positions, the coalesced way (16 lanes x 16 B = one 256-byte head row;
  // thread (rg, pq4) holds quad pq4 of positions rg + 32 i): requested behind the first two weight rounds, so that they
  // arrive under the QKV FMAs instead of costing two dependent round trips after them (persistent mode: BEFORE the wait --
  // the rows were written by this same role in earlier steps).  Positions past the live range re-read slot 0 of the row
  // (always valid; masked where consumed).
  const int npast = len - 1;                       // cached positions; the new token attends to itself from LDS
  const int rg = tid >> 4, pq4 = (tid & 15) * 4;
  const int* tb = a.tabs + (size_t)step_par * a.lay.S * a.Lmax + r * a.Lmax;
  auto slot_of = [&](int p) { if constexpr (PS) return p * a.lay.S + r; else return tb[p]; };
  int slot_new = 0;
  const Buf16 kbuf(a.Kc), vbuf(a.Vc);              // (persistent mode: the cache rows were written through -> sc1 loads)
  float4 kc[PSL], vc[PSL];
  auto load_cache_tile = [&]() {
    int slot[PSL];
#pragma unroll
    for (int i = 0; i < PSL; i++) slot[i] = slot_of(rg + 32 * i < npast ? rg + 32 * i : 0);
    slot_new = slot_of(npast);
    if constexpr (!PS) {
      if (npast > PT)                              // only the tail tiles of long sequences walk the table through LDS
        for (int p = tid; p < npast; p += NT) tbs[p] = tb[p];
    }
#pragma unroll
    for (int i = 0; i < PSL; i++) kc[i] = ld_f4<PS>(a.Kc, kbuf, (uint32_t)(slot[i] * d + h * 64 + pq4));
#pragma unroll
    for (int i = 0; i < PSL; i++) vc[i] = ld_f4<PS>(a.Vc, vbuf, (uint32_t)(slot[i] * d + h * 64 + pq4));
  };
  {
    // x + (bias + partial planes), s ascending (mod.rs:346-348): one element per thread, all planes in flight together
    const int c = tid < d ? tid : 0;
    const float* pp = a.pend + (int64_t)r * d + c;
    const int64_t plane = (int64_t)a.S * d;
    auto load_weights = [&]() {
      g_own = gld(a.ln_g + own_col); b_own = gld(a.ln_b + own_col);
      load_round(wr[0], 0);
      if (NIT > 1) load_round(wr[1], 1);
      if constexpr (PS) { if (tid < 192) qbias = gld(a.bqkv + (tid >> 6) * d + h * 64 + (tid & 63)); }
    };
    if constexpr (PS) {
      // persistent mode: the first weight rounds are in flight (or landed) while the block waits; the wait is a PRE-wake
      // (the stage before the producers has finished) -- the planes themselves arrive as tagged granules, re-read until
      // every tag is the producers' (arrival and payload in one round trip)
#if !defined(WB_PS_DRY)
      if (!ps.known_dead) {
        load_weights();
        load_cache_tile();
      }
#endif
      if (!ps_wait(ps)) return false;
      dead = ps.known_dead ? 1 : ld_i<true>(ps.dead + r);
      if (r >= n_rows || dead) { ps.saw_dead = true; return true; }
      const Buf16 xgb(a.g_x_in), pgb(a.g_pend);
      const uint32_t pvo = (uint32_t)(r * d + c);
      const int iplane = a.S * d;
      if (a.KSp > 0) ps_nap<2>();                   // (the previous layer's MLP has only just started)
      constexpr int FP = 4 * DPL;                   // the 4 d / 64 planes of the previous layer's MLP
      float v = 0.f, accp = a.KSp > 0 ? gld(a.pbias + c) : 0.f;
      unsigned sweeps = 0;
      // (a granule that has arrived is not read again: the retries of the last planes are short round trips)
      Gran gx{0u, 0.f}, g[FP];
#pragma unroll
      for (int j = 0; j < FP; j++) g[j] = Gran{0u, 0.f};
      for (;;) {
        if (gx.tag != ps.tag_in) gx = ld_gran(xgb, pvo, 0u);
        if (a.KSp > 0) {
#pragma unroll
          for (int j = 0; j < FP; j++)
            if (j < a.KSp && g[j].tag != ps.tag_in) g[j] = ld_gran(pgb, pvo, (uint32_t)(j * iplane));
        }
        bool ok = gx.tag == ps.tag_in;
        if (a.KSp > 0) {
#pragma unroll
          for (int j = 0; j < FP; j++) ok &= (j >= a.KSp) || g[j].tag == ps.tag_in;
        }
        if (ok) {
          v = gx.v;
          if (a.KSp > 0) {
#pragma unroll
            for (int j = 0; j < FP; j++) accp += (j < a.KSp) ? g[j].v : 0.f;       // s ascending (mod.rs:346-348)
            v += accp;
          }
          break;
        }
        if (ps_sweep_retry(sweeps, ps)) { *ps.lds_flag = 0; break; }
      }
      xfold = v;
      if (tid < d) hs[tid] = v;
    } else {
    float v = a.x_in[(int64_t)r * d + c];
    constexpr int FP = 32;                          // planes per round: 4 d / 64 <= 32 MLP planes in ONE round trip
    float t[FP];
    float accp = 0.f;
    if (a.KSp > 0) {
      accp = gld(a.pbias + c);
#pragma unroll
      for (int j = 0; j < FP; j++) t[j] = pp[(int64_t)min(j, a.KSp - 1) * plane];
    }
    load_weights();
    if (r >= n_rows || dead) return true;          // (block-uniform; the first wait of the kernel)
    if (a.KSp > 0) {
#pragma unroll
      for (int j = 0; j < FP; j++) accp += (j < a.KSp) ? t[j] : 0.f;
      for (int sp = FP; sp < a.KSp; sp += FP) {
#pragma unroll
        for (int j = 0; j < FP; j++) t[j] = pp[(int64_t)min(sp + j, a.KSp - 1) * plane];
#pragma unroll
        for (int j = 0; j < FP; j++) accp += (sp + j < a.KSp) ? t[j] : 0.f;
      }
      v += accp;
    }
    xfold = v;
    if (tid < d) hs[tid] = v;
    }
  }
  if constexpr (!PS) qbias = tid < 192 ? a.bqkv[(tid >> 6) * d + h * 64 + (tid & 63)] : 0.f;   // key part is zero (mod.rs:402-404)
  if constexpr (!PS) load_cache_tile();
  WB_STAMP(1);
  if constexpr (PS) ps_stamp(ps, 2);
  __syncthreads();
  if constexpr (PS) { if (!ps_sweeps_ok(ps)) return false; }
#if defined(WB_PS_DRY)
  if constexpr (PS) {
    if (tid < d / 4) { const Buf16 pgo(a.g_P); st_gran4(pgo, (uint32_t)((h * a.S + r) * d + tid * 4), ps.tag_out, make_float4(0.f, 0.f, 0.f, 0.f)); }
    if (h == 0 && tid < d) { const Buf16 xgo(a.g_x_out); st_gran(xgo, (uint32_t)(r * d + tid), ps.tag_out, xfold); }
    return true;
  }
#endif
  // LayerNorm: every wave takes the row's statistics itself and normalises its own KW rows into a register (lane l: row
  // wave KW + l) -- no second barrier; the QKV loop broadcasts them with v_readlane
  float xn_own;
  {
    float mean, denom;
    ln_stats_lds<DPL>(hs, d, lane, a.ln_eps, a.ln_inside, mean, denom);
    xn_own = (hs[own_col] - mean) / denom * g_own + b_own;
  }
  if constexpr (PS) ps_stamp_fine(ps, 4);
  WB_STAMP(2);
  // ---- QKV for head h (rolled on purpose: unrolled, every round's loads are hoisted to the top and spill)
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int it = 0; it < NIT; it += 2) {
#pragma unroll
    for (int b = 0; b < 2; b++) {
      if (it + b < NIT) {
        const int kb = RK * (it + b);              // (row of the wave's own KW = lane that holds it)
#pragma unroll
        for (int j = 0; j < RK; j++) {
          const float xv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xn_own), kb + j));
          acc[0] += xv * wr[b][j].x; acc[1] += xv * wr[b][j].y; acc[2] += xv * wr[b][j].z; acc[3] += xv * wr[b][j].w;
        }
        if (it + b + 2 < NIT) load_round(wr[b], it + b + 2);
      }
    }
  }
  WB_STAMP(3);
  if constexpr (PS) ps_stamp(ps, 3);
  __syncthreads();   // (also pins the loads below behind the FMAs: the weight registers are free now)
  // ---- requested now: the Wo slice (consumed last)
  const int cf = tid % CF, jg = tid / CF;
  const bool p5 = jg < G;
  const int jb = (p5 ? jg : 0) * RPG;
  float4 wo[RPG];
  {
    const float* wp = a.Wo + (int64_t)(h * 64 + jb) * d + cf * 4;
#pragma unroll
    for (int i = 0; i < RPG; i++) wo[i] = ld_w4(wp + (int64_t)i * d);
  }
  if (seg_ok) *reinterpret_cast<float4*>(&red[wave * 192 + seg * 64 + c4]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
  if (tid < 192) {
    float v = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < 8; w8++) v += red[w8 * 192 + tid];            // wave order fixed
    v += qbias;
    if (tid < 128) v *= a.scale;                                        // q * s, k * s  (mod.rs:506-514)
    qkv[tid] = v;
  }
  __syncthreads();
  if constexpr (PS) ps_stamp_fine(ps, 5);
  WB_STAMP(4);
  // ---- scores: 4-term partial dots summed over the 16 lanes of a position's row (DPP); tail tiles (sequences longer
  // than 128 cached positions) reload through the LDS copy of the table
  {
    const float4 q4 = *reinterpret_cast<const float4*>(&qkv[pq4]);
    auto row_dot = [&](const float4& k4) {
      float s = q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w;
      s += dpp_f<DPP_QUAD_XOR1, 0xF>(0.f, s);
      s += dpp_f<DPP_QUAD_XOR2, 0xF>(0.f, s);
      s += dpp_f<DPP_ROW_HALF_MIRROR, 0xF>(0.f, s);
      s += dpp_f<DPP_ROW_MIRROR, 0xF>(0.f, s);
      return s;
    };
#pragma unroll
    for (int i = 0; i < PSL; i++) {
      const int p = rg + 32 * i;
      const float s = row_dot(kc[i]);
      if ((tid & 15) == 0 && p < npast) sc[p] = s;
    }
    for (int t0 = PT; t0 < npast; t0 += PT) {
      float4 kt[PSL];
#pragma unroll
      for (int i = 0; i < PSL; i++) {
        const int p = min(t0 + rg + 32 * i, npast - 1);
        int sl;
        if constexpr (PS) sl = p * a.lay.S + r; else sl = tbs[p];
        kt[i] = ld_f4<PS>(a.Kc, kbuf, (uint32_t)(sl * d + h * 64 + pq4));
      }
#pragma unroll
      for (int i = 0; i < PSL; i++) {
        const int p = t0 + rg + 32 * i;
        const float s = row_dot(kt[i]);
        if ((tid & 15) == 0 && p < npast) sc[p] = s;
      }
    }
    if (wave == 7) {                                                    // the new token's own key (k * s from LDS)
      const float s = wave_sum(qkv[lane] * qkv[64 + lane]);
      if (lane == 0) sc[npast] = s;
    }
  }
  __syncthreads();
  WB_STAMP(5);
  if constexpr (PS) ps_stamp(ps, 4);
  // softmax statistics, redundantly per wave (no extra barrier): m, l over all positions
  float m = -INFINITY;
  for (int p = lane; p < len; p += 64) m = fmaxf(m, sc[p]);
  m = wave_max(m);
  float l = 0.f;
  for (int p = lane; p < len; p += 64) l += expf(sc[p] - m);
  l = wave_sum(l);
  {
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < PSL; i++) {
      const int p = rg + 32 * i;
      const float pk = p < npast ? expf(sc[p] - m) : 0.f;
      o.x += pk * vc[i].x; o.y += pk * vc[i].y; o.z += pk * vc[i].z; o.w += pk * vc[i].w;
    }
    for (int t0 = PT; t0 < npast; t0 += PT) {
      float4 vt[PSL];
#pragma unroll
      for (int i = 0; i < PSL; i++) {
        const int p = min(t0 + rg + 32 * i, npast - 1);
        int sl;
        if constexpr (PS) sl = p * a.lay.S + r; else sl = tbs[p];
        vt[i] = ld_f4<PS>(a.Vc, vbuf, (uint32_t)(sl * d + h * 64 + pq4));
      }
#pragma unroll
      for (int i = 0; i < PSL; i++) {
        const int p = t0 + rg + 32 * i;
        const float pk = p < npast ? expf(sc[p] - m) : 0.f;
        o.x += pk * vt[i].x; o.y += pk * vt[i].y; o.z += pk * vt[i].z; o.w += pk * vt[i].w;
      }
    }
    *reinterpret_cast<float4*>(&part[rg][pq4]) = o;
  }
  __syncthreads();
  if (tid < 64) {
    float v = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < 32; g2++) v += part[g2][tid];                 // row-group order fixed
    v += expf(sc[npast] - m) * qkv[128 + tid];                          // the new token's own value
    att[tid] = v / l;
  }
  __syncthreads();
  WB_STAMP(6);
  if constexpr (PS) ps_stamp(ps, 5);
  // ---- plane h, row r = att Wo[head h rows, :]
  float ov[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < RPG; i++) {
    const float av = att[jb + i];
    ov[0] += av * wo[i].x; ov[1] += av * wo[i].y; ov[2] += av * wo[i].z; ov[3] += av * wo[i].w;
  }
  if (p5 && jg > 0)                                 // (the QKV partials in `red` are dead)
    *reinterpret_cast<float4*>(&red[(jg - 1) * d + cf * 4]) = make_float4(ov[0], ov[1], ov[2], ov[3]);
  __syncthreads();
  if (jg == 0) {
#pragma unroll
    for (int g2 = 1; g2 < G; g2++) {
      const float4 t = *reinterpret_cast<const float4*>(&red[(g2 - 1) * d + cf * 4]);
      ov[0] += t.x; ov[1] += t.y; ov[2] += t.z; ov[3] += t.w;
    }
    if constexpr (PS) {
      const Buf16 pgo(a.g_P);
      st_gran4(pgo, (uint32_t)((h * a.S + r) * d + cf * 4), ps.tag_out, make_float4(ov[0], ov[1], ov[2], ov[3]));
    } else {
      *reinterpret_cast<float4*>(&a.P[((int64_t)h * a.S + r) * d + cf * 4]) = make_float4(ov[0], ov[1], ov[2], ov[3]);
    }
  }
  // ---- the stores that are not on anybody's critical path: k * s, v of the new token into the cache, the folded stream
  if (tid >= 64 && tid < 192) {
    float* dst = (tid < 128 ? a.Kc : a.Vc) + (int64_t)slot_new * d + h * 64 + (tid & 63);
    st_f<PS>(dst, qkv[tid]);
  }
  if (h == 0 && tid < d) {
    if constexpr (PS) { const Buf16 xgo(a.g_x_out); st_gran(xgo, (uint32_t)(r * d + tid), ps.tag_out, xfold); }
    else a.x_out[(int64_t)r * d + tid] = xfold;
  }
  WB_STAMP(7);
  WB_STAMP_FLUSH(a, 8);
  return true;
}

// ---------------------------------------------------------------------------------------------------------
// Cross-attention block.  block = 512 threads, grid = (8, rows): block (h, r) owns head h of beam r (x = head keeps a
// head's blocks on one XCD: the Wq / Wo slices and, for beams of the same window, the cached K/V cross the fabric once).
//   LN(x + pending) -> q = . Wq[:, head h] + bq, * s -> scores against ALL of the window's cached K (C <= 768 keys)
//   -> softmax -> . V -> . Wo[head h rows, :] -> plane h of [H][S][d]                              (mod.rs:482-490)
// One launch instead of cross-attention (per 128-key chunk) + chunk combine + out-projection GEMV.
//
// A block streams ~0.6 MB (K and V of its head: 2 x C x 256 B, the Wq and Wo slices: 2 x d x 256 B) through ONE CU, so
// the kernel is organised around bytes in flight.  K/V rows are read the coalesced way (16 lanes x 16 B = one 256-byte
// head row, 32 rows per wave-instruction round) into a register ring that holds the WHOLE K of the head (6 tiles x 128
// keys, 24 float4 per thread); every K register is refilled with the V row of the same key as soon as its score is
// done, so the V stream is in flight under the softmax statistics.  Scores are 4-term partial dots reduced over the
// 16 lanes of a row by DPP (no LDS transpose); the output is a float4 of partial sums per thread, reduced over the 32
// row groups in a fixed order.  The Wq slice arrives under the fold + LayerNorm, the Wo slice under the scores.
//
// NP = 2 (768 < C <= 1536 keys per window: the opt-in 30 s window geometry, C = 1500): the ring still holds 768 keys, so
// FOUR streams pass through it -- K of keys [0, 768), K of keys [768, C) (a register is refilled as soon as its score
// is done), V of keys [0, 768) (refilled under the second pass of scores), V of keys [768, C) (refilled as the first
// pass of the output sum consumes a register).  Straight-line code: a window with C <= 768 keys (the tail window) makes
// the same passes over clamped rows, its scores past C never stored and their probabilities zero.
template <int DPL, bool PS, int NP = 1>
__device__ __forceinline__ bool dec_cross_body(const CrossFusedArgs& a, const int h, const int r, const PsStep& ps) {
  constexpr int NT = 512;
  constexpr int PASS_C = CROSS_FUSED_MAX_C, CMAX = NP * PASS_C;
  constexpr int d = 64 * DPL;
  constexpr int NWQ = d / 32;                      // Wq rows per thread
  constexpr int CF = d / 4;
  constexpr int G = CF <= 32 ? 8 : 4;
  constexpr int RPG = 64 / G;
  constexpr int KT = 128, NTILE = CROSS_FUSED_MAX_C / KT, SL = KT * 16 / NT;   // 6 tiles x 4 float4 slots per thread
  static_assert(CROSS_FUSED_MAX_C % KT == 0 && SL * NT == KT * 16, "tile geometry");
  __shared__ __attribute__((aligned(16))) float hs[d];
  __shared__ __attribute__((aligned(16))) float red[8][64];
  __shared__ __attribute__((aligned(16))) float qv[64];
  __shared__ float sc[CMAX];
  __shared__ float pbuf[CMAX];
  __shared__ __attribute__((aligned(16))) float part[32][64];
  __shared__ __attribute__((aligned(16))) float att[64];
  __shared__ __attribute__((aligned(16))) float obuf[(G - 1) * d];
  const int tid = role_tid<PS>(), lane = tid & 63, wave = tid >> 6;
  if (h >= a.n_head) return true;
  int n_live, w_row, dead;
  if constexpr (PS) { n_live = ps.n_rows; w_row = r; dead = 0; }     // (one beam per window: slot == window)
  else {
    n_live = a.st[ST_N]; w_row = a.st[a.lay.win + r];
    dead = a.st[a.lay.dead + r];                   // chained greedy decode: this row's window has already ended
  }
  // the window geometry of the first 8 windows rides with the state words -- as VECTOR loads (lane i holds window i; a
  // scalar load here would stall the next kernel-argument wait, lgkmcnt being one counter): the K stream can then start
  // one round trip after kernel entry instead of two (row -> window -> geometry)
  const int wi8 = min(lane & 7, a.lay.W - 1);
  const int vC8 = gld(a.win_C + wi8), vR8 = gld(a.win_row0 + wi8);
  // ---- requested first (in order of use): fold operands, LayerNorm parameters, bias, the Wq slice
  float xfold;
  const int rg = tid >> 4, c4 = (tid & 15) * 4;
  // LayerNorm parameters of the 4 NWQ columns this wave's Wq rows need: lane l < 4 NWQ owns column 4 wave + (l & 3) +
  // 32 (l >> 2) = row rg + 32 i of thread group (rg = 4 wave + (l & 3), i = l >> 2)
  constexpr int NOWN = 4 * NWQ;
  const int own_col = 4 * wave + (lane & 3) + 32 * ((lane < NOWN ? lane : 0) >> 2);
  float g_own = 0.f, b_own = 0.f;
  float4 wqr[NWQ];
  float4 kv[NTILE][SL];
  const float* Kh; const float* Vh; int C;
  float qbias = 0.f;
  auto load_weights = [&]() {
    g_own = gld(a.ln_g + own_col); b_own = gld(a.ln_b + own_col);
    if constexpr (PS) { if (tid < 64) qbias = gld(a.bq + h * 64 + tid); }
    {
      const float* wp = a.Wq + (int64_t)rg * d + h * 64 + c4;
#pragma unroll
      for (int i = 0; i < NWQ; i++) wqr[i] = ld_w4(wp + (int64_t)(32 * i) * d);
    }
  };
  // the head's cached K, all of it, into the register ring (key = tile * 128 + rg + 32 * slot; quad c4)
  auto load_keys = [&]() {
    int C_w, row0_w;
    {
      const int ws = __builtin_amdgcn_readfirstlane(w_row);
      if (ws < 8) { C_w = __builtin_amdgcn_readlane(vC8, ws); row0_w = __builtin_amdgcn_readlane(vR8, ws); }
      else { C_w = gld(a.win_C + ws); row0_w = gld(a.win_row0 + ws); }
    }
    C = min(C_w, CMAX);
    // uniform base + 32-bit per-lane offsets; keys past C re-read row C - 1 (their scores are never stored and their
    // probabilities are zero), so every load is unconditional: no predicate sits between two requests
    Kh = a.ckv + (int64_t)row0_w * a.ldkv + a.koff + h * 64;                  // K pre-scaled at projection time
    Vh = Kh + d;
#pragma unroll
    for (int t = 0; t < NTILE; t++)
#pragma unroll
      for (int i = 0; i < SL; i++) {
        const int key = min(t * KT + rg + 32 * i, C - 1);
        kv[t][i] = gld4(Kh + (key * a.ldkv + c4));
      }
  };
  {
    const int c = tid < d ? tid : 0;
    const float* pp = a.pend + (int64_t)r * d + c;
    const int64_t plane = (int64_t)a.S * d;
    if constexpr (PS) {
      // persistent mode: neither the weights nor the window's cached keys depend on the predecessor -- the Wq slice and
      // the WHOLE K of the head are in flight (or landed) while the block waits (pre-wake); the self-attention planes
      // arrive as tagged granules
#if !defined(WB_PS_DRY)
      if (!ps.known_dead) {
        load_weights();
        load_keys();
      }
#endif
      if (!ps_wait(ps)) return false;
      dead = ps.known_dead ? 1 : ld_i<true>(ps.dead + r);
      if (r >= n_live || dead) { ps.saw_dead = true; return true; }
      const Buf16 xgb(a.g_x_in), pgb(a.g_pend);
      const uint32_t pvo = (uint32_t)(r * d + c);
      const int iplane = a.S * d;
      ps_nap<4>();                                  // (the self-attention blocks have only just started)
      constexpr int FP = 8;                         // <= 8 head planes
      float v = 0.f, accp = gld(a.pbias + c);
      unsigned sweeps = 0;
      Gran gx{0u, 0.f}, g[FP];
#pragma unroll
      for (int j = 0; j < FP; j++) g[j] = Gran{0u, 0.f};
      for (;;) {
        if (gx.tag != ps.tag_in) gx = ld_gran(xgb, pvo, 0u);
#pragma unroll
        for (int j = 0; j < FP; j++)
          if (j < a.KSp && g[j].tag != ps.tag_in) g[j] = ld_gran(pgb, pvo, (uint32_t)(j * iplane));
        bool ok = gx.tag == ps.tag_in;
#pragma unroll
        for (int j = 0; j < FP; j++) ok &= (j >= a.KSp) || g[j].tag == ps.tag_in;
        if (ok) {
#pragma unroll
          for (int j = 0; j < FP; j++) accp += (j < a.KSp) ? g[j].v : 0.f;         // s ascending (mod.rs:346-348)
          v = gx.v + accp;
          break;
        }
        if (ps_sweep_retry(sweeps, ps)) { *ps.lds_flag = 0; break; }
      }
      xfold = v;
      if (tid < d) hs[tid] = v;
    } else {
    float v = a.x_in[(int64_t)r * d + c];
    constexpr int FP = 16;
    float t[FP];
    float accp = 0.f;
    if (a.KSp > 0) {
      accp = gld(a.pbias + c);
#pragma unroll
      for (int j = 0; j < FP; j++) t[j] = pp[(int64_t)min(j, a.KSp - 1) * plane];
    }
    load_weights();
    if (r >= n_live || dead) return true;          // (block-uniform; the first wait of the kernel)
    if (a.KSp > 0) {
#pragma unroll
      for (int j = 0; j < FP; j++) accp += (j < a.KSp) ? t[j] : 0.f;
      for (int sp = FP; sp < a.KSp; sp += FP) {
#pragma unroll
        for (int j = 0; j < FP; j++) t[j] = pp[(int64_t)min(sp + j, a.KSp - 1) * plane];
#pragma unroll
        for (int j = 0; j < FP; j++) accp += (sp + j < a.KSp) ? t[j] : 0.f;
      }
      v += accp;                                   // x + (bias + partials), s ascending  (mod.rs:346-348)
    }
    xfold = v;
    if (tid < d) hs[tid] = v;
    }
  }
  if constexpr (!PS) qbias = tid < 64 ? a.bq[h * 64 + tid] : 0.f;
  if constexpr (!PS) load_keys();
  if constexpr (PS) ps_stamp(ps, 2);
  __syncthreads();
  if constexpr (PS) { if (!ps_sweeps_ok(ps)) return false; }
#if defined(WB_PS_DRY)
  if constexpr (PS) {
    if (tid < d / 4) { const Buf16 pgo(a.g_P); st_gran4(pgo, (uint32_t)((h * a.S + r) * d + tid * 4), ps.tag_out, make_float4(0.f, 0.f, 0.f, 0.f)); }
    if (h == 0 && tid < d) { const Buf16 xgo(a.g_x_out); st_gran(xgo, (uint32_t)(r * d + tid), ps.tag_out, xfold); }
    return true;
  }
#endif
  // LayerNorm: statistics per wave, each wave normalises the columns its own Wq rows multiply -- no second barrier
  float xn_own;
  {
    float mean, denom;
    ln_stats_lds<DPL>(hs, d, lane, a.ln_eps, a.ln_inside, mean, denom);
    xn_own = (hs[own_col] - mean) / denom * g_own + b_own;
  }
  if constexpr (PS) ps_stamp_fine(ps, 4);
  // ---- q = (cross_attn_ln(x) Wq + bq) * s for head h  (mod.rs:483, :506-509)
  {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int q4 = lane >> 4;
#pragma unroll
    for (int i = 0; i < NWQ; i++) {
      const float xv = lane_get(xn_own, 4 * i + q4);
      acc[0] += xv * wqr[i].x; acc[1] += xv * wqr[i].y; acc[2] += xv * wqr[i].z; acc[3] += xv * wqr[i].w;
    }
#pragma unroll
    for (int c = 0; c < 4; c++) { acc[c] = xor16_sum(acc[c]); acc[c] = xor32_sum(acc[c]); }
    if (lane < 16) *reinterpret_cast<float4*>(&red[wave][c4]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
  __syncthreads();   // (also keeps the loads below behind the Wq registers' last use)
  if constexpr (PS) ps_stamp_fine(ps, 5);
  // ---- requested now: the Wo slice (consumed last)
  const int cf = tid % CF, jg = tid / CF;
  const bool p5 = jg < G;
  const int jb = (p5 ? jg : 0) * RPG;
  float4 wo[RPG];
  {
    const float* wp = a.Wo + (int64_t)(h * 64 + jb) * d + cf * 4;
#pragma unroll
    for (int i = 0; i < RPG; i++) wo[i] = ld_w4(wp + (int64_t)i * d);
  }
  if (tid < 64) {
    float v = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < 8; w8++) v += red[w8][tid];                   // wave order fixed
    qv[tid] = (v + qbias) * a.scale;
  }
  __syncthreads();
  if constexpr (PS) ps_stamp(ps, 3);
  // ---- scores: partial dot over this thread's quad, summed over the 16 lanes of the key row; the register then
  // takes the V row of the same key
  {
    const float4 q4 = *reinterpret_cast<const float4*>(&qv[c4]);
#pragma unroll
    for (int p = 0; p < NP; p++) {
      // what refills a register once its score is done: the next pass's K row, after the last pass the V row of pass 0
      const float* nxt = p + 1 < NP ? Kh + (p + 1) * PASS_C * a.ldkv : Vh;
      int rgp = rg;
      if constexpr (NP > 1) WB_LAUNDER_V(rgp);       // (a pass's row offsets are computed in that pass, not hoisted and spilled)
#pragma unroll
      for (int t = 0; t < NTILE; t++) {
        // (two passes: the scheduler must not lift a tile's refills above the previous tile's scores -- the ring is the
        // register file)
        if constexpr (NP > 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < SL; i++) {
          const int k0 = t * KT + rgp + 32 * i, key = p * PASS_C + k0;
          float s = q4.x * kv[t][i].x + q4.y * kv[t][i].y + q4.z * kv[t][i].z + q4.w * kv[t][i].w;
          s += dpp_f<DPP_QUAD_XOR1, 0xF>(0.f, s);
          s += dpp_f<DPP_QUAD_XOR2, 0xF>(0.f, s);
          s += dpp_f<DPP_ROW_HALF_MIRROR, 0xF>(0.f, s);
          s += dpp_f<DPP_ROW_MIRROR, 0xF>(0.f, s);
          if ((tid & 15) == 0 && key < C) sc[key] = s;
          const int lim = p + 1 < NP ? max(C - 1 - (p + 1) * PASS_C, -(p + 1) * PASS_C) : C - 1;   // (clamped rows are never consumed)
          kv[t][i] = gld4(nxt + (min(k0, lim) * a.ldkv + c4));
        }
      }
    }
  }
  __syncthreads();
  if constexpr (PS) ps_stamp(ps, 4);
  // softmax statistics, redundantly per wave (no extra barrier); then the probabilities, two keys per thread
  float m = -INFINITY;
  for (int j = lane; j < C; j += 64) m = fmaxf(m, sc[j]);
  m = wave_max(m);
  float l = 0.f;
  for (int j = lane; j < C; j += 64) l += expf(sc[j] - m);
  l = wave_sum(l);
  for (int j = tid; j < C; j += NT) pbuf[j] = expf(sc[j] - m);
  __syncthreads();
  // ---- o[c4 .. c4 + 3] partial over this thread's keys (tile, slot ascending), then over the 32 row groups
  {
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int p = 0; p < NP; p++) {
      int rgp = rg;
      if constexpr (NP > 1) WB_LAUNDER_V(rgp);
#pragma unroll
      for (int t = 0; t < NTILE; t++) {
        // (two passes: a tile's refills stay behind the sums that consume its registers -- the sums have no side effect,
        // so without the pin every refill of the pass is requested first and the ring doubles)
        if constexpr (NP > 1) WB_PIN_F4(o);
#pragma unroll
        for (int i = 0; i < SL; i++) {
          const int k0 = t * KT + rgp + 32 * i, key = p * PASS_C + k0;
          const float pk = key < C ? pbuf[key] : 0.f;
          o.x += pk * kv[t][i].x; o.y += pk * kv[t][i].y; o.z += pk * kv[t][i].z; o.w += pk * kv[t][i].w;
          if (p + 1 < NP)                           // the register takes the V row of the next pass's key
            kv[t][i] = gld4(Vh + (min(key + PASS_C, C - 1) * a.ldkv + c4));
        }
      }
    }
    *reinterpret_cast<float4*>(&part[rg][c4]) = o;
  }
  __syncthreads();
  if (tid < 64) {
    float v = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < 32; g2++) v += part[g2][tid];                 // row-group order fixed
    att[tid] = v / l;
  }
  __syncthreads();
  if constexpr (PS) ps_stamp(ps, 5);
  // ---- plane h, row r = att Wo[head h rows, :]
  float ov[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < RPG; i++) {
    const float av = att[jb + i];
    ov[0] += av * wo[i].x; ov[1] += av * wo[i].y; ov[2] += av * wo[i].z; ov[3] += av * wo[i].w;
  }
  if (p5 && jg > 0) *reinterpret_cast<float4*>(&obuf[(jg - 1) * d + cf * 4]) = make_float4(ov[0], ov[1], ov[2], ov[3]);
  __syncthreads();
  if (jg == 0) {
#pragma unroll
    for (int g2 = 1; g2 < G; g2++) {
      const float4 t = *reinterpret_cast<const float4*>(&obuf[(g2 - 1) * d + cf * 4]);
      ov[0] += t.x; ov[1] += t.y; ov[2] += t.z; ov[3] += t.w;
    }
    if constexpr (PS) {
      const Buf16 pgo(a.g_P);
      st_gran4(pgo, (uint32_t)((h * a.S + r) * d + cf * 4), ps.tag_out, make_float4(ov[0], ov[1], ov[2], ov[3]));
    } else {
      *reinterpret_cast<float4*>(&a.P[((int64_t)h * a.S + r) * d + cf * 4]) = make_float4(ov[0], ov[1], ov[2], ov[3]);
    }
  }
  if (h == 0 && tid < d) {                         // the folded stream, off the critical path
    if constexpr (PS) { const Buf16 xgo(a.g_x_out); st_gran(xgo, (uint32_t)(r * d + tid), ps.tag_out, xfold); }
    else a.x_out[(int64_t)r * d + tid] = xfold;
  }
  return true;
}


}  // namespace fused
}  // namespace wb
