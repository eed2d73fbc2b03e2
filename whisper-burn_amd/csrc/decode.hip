// K7/K8: KV-cached single-token decode step (HBM-bound weight / KV streaming, latency-critical).
//
// The reference re-runs the whole decoder over the whole prefix for every beam at every step
// and re-projects the cross-attention K/V in every layer (/root/reference/src/transcribe.rs:270,
// src/model/mod.rs:482-490), then ships [n, L, V] logits to log_softmax and V floats per beam to
// the host (transcribe.rs:276-284).  Here a step touches each decoder weight once, in 7-8 launches
// per layer + 2-3 (small models project the cross-attention queries inside the attention blocks; in the
// device-chained greedy loop the merge kernel prepares the next step):
//   skinny GEMMs (rows = live beams) stream W[K][N] once with K split over blocks into
//   deterministic partial sums; the CONSUMER folds the partials in its prologue (+bias, residual,
//   LayerNorm / GELU / q-scale / attention-chunk combine), so there is no reduction kernel and
//   no atomic; weight rounds are software-pipelined through 2-3 register sets, the first one
//   requested ahead of the prologue so the two global latencies overlap;
//   self-attention reads a paged self-KV cache through per-beam position tables (beam
//   re-indexing = copying a row of ints); cross-attention streams each window's cached K/V once
//   for all of that window's beams, split over key chunks (flash-decoding);
//   the tied-embedding logits kernel streams E^T [d][V] and leaves per-128-column-tile
//   max / sum-exp / top-k, merged by one small block per beam that writes the k (id, log-prob)
//   pairs straight into mapped host memory.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "decode.h"
#include "wave_ops.h"

namespace wb {
namespace {

constexpr int GV_KSL_MAX = 512;   // K-slice capacity of the non-LN prologues
constexpr int DMAX = 1280;        // largest n_state (large-v2); LN prologues keep whole rows in LDS

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// bias[c] + sum_s P[s][row][c] with the loads issued together (fixed summation order)
__device__ __forceinline__ float fold_partials(const float* __restrict__ P, int KS, int64_t plane, int64_t off,
                                               float init) {
  float v[KS_MAX];     // (uniform predicate: scalar branches, the loads stay in flight together)
#pragma unroll
  for (int s = 0; s < KS_MAX; s++) v[s] = (s < KS) ? P[(int64_t)s * plane + off] : 0.f;
  float acc = init;
#pragma unroll
  for (int s = 0; s < KS_MAX; s++) acc += v[s];
  return acc;
}

// flash-decoding combine of the cross-attention key-chunk partials {m, l, o[64]} of one (row, head):
// every load is issued before the first use (n_chunks is uniform: scalar branches around the loads)
__device__ __forceinline__ float combine_chunks(const float* __restrict__ ca, int n_chunks, int dh) {
  float m[CA_NCH_MAX], l[CA_NCH_MAX], o[CA_NCH_MAX];
#pragma unroll
  for (int c = 0; c < CA_NCH_MAX; c++) {
    const float* p = ca + c * CA_STRIDE;
    m[c] = -1.0e30f; l[c] = 0.f; o[c] = 0.f;
    if (c < n_chunks) { m[c] = p[0]; l[c] = p[1]; o[c] = p[2 + dh]; }
  }
  float M = -1.0e30f;
#pragma unroll
  for (int c = 0; c < CA_NCH_MAX; c++) M = fmaxf(M, m[c]);
  float num = 0.f, den = 0.f;
#pragma unroll
  for (int c = 0; c < CA_NCH_MAX; c++) {
    const float wgt = c < n_chunks ? expf(m[c] - M) : 0.f;
    num += wgt * o[c];
    den += wgt * l[c];
  }
  return num / den;
}

// ---- step prepare: state host->device, position tables, token/position embedding (mod.rs:141-146) ----
__global__ void dec_prepare_kernel(const int* __restrict__ st_host, int* __restrict__ st_dev, StepLayout lay,
                                   int* __restrict__ tabs, int Lmax, const float* __restrict__ E,
                                   const float* __restrict__ pos, int d, float* __restrict__ x,
                                   const int* __restrict__ gctl) {
  const int i = blockIdx.x;
  int n, step, len, parent, tok;
  if (gctl) {
    // device-chained greedy decode: the step state is derived on the device from the control block
    // the previous step's merge kernel left behind (one beam per window, slot == window)
    n = lay.W; step = gctl[GC_STEP]; len = step + 1; parent = i; tok = i < n ? gctl[GC_HDR + i] : 0;
    if (i == 0)
      for (int e = threadIdx.x; e < lay.total; e += blockDim.x) {
        int v = 0;
        if (e == ST_N) v = n;
        else if (e == ST_STEP) v = step;
        else if (e >= lay.tok && e < lay.tok + n) v = gctl[GC_HDR + (e - lay.tok)];
        else if (e >= lay.parent && e < lay.parent + n) v = e - lay.parent;
        else if (e >= lay.len && e < lay.len + n) v = len;
        else if (e >= lay.win && e < lay.win + n) v = e - lay.win;
        else if (e >= lay.dead && e < lay.dead + n) v = gctl[GC_HDR + lay.S + (e - lay.dead)];
        else if (e >= lay.win_nb && e < lay.win_nb + n) v = gctl[GC_HDR + lay.S + (e - lay.win_nb)] ? 0 : 1;   // a finished window streams no K/V
        else if (e >= lay.win_slots && (e - lay.win_slots) % MAX_BEAMS == 0 && (e - lay.win_slots) / MAX_BEAMS < n)
          v = (e - lay.win_slots) / MAX_BEAMS;
        st_dev[e] = v;
      }
  } else {
    if (i == 0)
      for (int e = threadIdx.x; e < lay.total; e += blockDim.x) st_dev[e] = st_host[e];
    n = st_host[ST_N]; step = st_host[ST_STEP];
    len = st_host[lay.len + i]; parent = st_host[lay.parent + i]; tok = st_host[lay.tok + i];
  }
  if (i >= n) return;
  int* tab_new = tabs + (size_t)(step & 1) * lay.S * Lmax;           // position tables are double-buffered by step parity
  const int* tab_old = tabs + (size_t)((step & 1) ^ 1) * lay.S * Lmax;
  if (parent >= 0)
    for (int p = threadIdx.x; p < len - 1; p += blockDim.x) tab_new[i * Lmax + p] = tab_old[parent * Lmax + p];
  if (threadIdx.x == 0) tab_new[i * Lmax + len - 1] = step * lay.S + i;
  const float4* e = reinterpret_cast<const float4*>(E + (int64_t)tok * d);
  const float4* pp = reinterpret_cast<const float4*>(pos + (int64_t)(len - 1) * d);
  float4* o = reinterpret_cast<float4*>(x + (int64_t)i * d);
  for (int c = threadIdx.x; c < (d >> 2); c += blockDim.x) {
    float4 a = e[c], b = pp[c];
    o[c] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

// ---- stand-alone residual resolve + LayerNorm (used when more than 8 beams are live) ------------
template <int VPT>
__global__ __launch_bounds__(256) void dec_resolve_ln_kernel(const int* __restrict__ st, const float* __restrict__ x_in,
                                                             float* __restrict__ x_out, const float* __restrict__ P,
                                                             int KS, int S, const float* __restrict__ bias, int d,
                                                             const float* __restrict__ g,
                                                             const float* __restrict__ b, float eps,
                                                             int eps_inside_sqrt, float* __restrict__ h) {
  __shared__ float red[8];
  const int r = blockIdx.x;
  const int tid = threadIdx.x;
  // One memory round trip: the live-row count, the row, every pending plane, the bias and the LayerNorm parameters are
  // requested before the first use of any of them, and the folded stream is stored last (the launch was 6.6 us for ~1.5 MB
  // at large-v2's 38 rows, three dependent round trips of it: profiles/r03_m_layer_cycle_large_v2.txt).  Columns past d
  // re-read column 0 and planes past KS are zero; neither is used.
  const int n_live = st[ST_N];
  float xv[VPT], gv[VPT], bv[VPT], biasv[VPT], t[VPT][KS_MAX];
  const int64_t plane = (int64_t)S * d;
#pragma unroll
  for (int i = 0; i < VPT; i++) {
    const int c = tid + i * 256, cc = c < d ? c : 0;     // (d = 128: threads past the row re-read column 0)
    xv[i] = x_in[(int64_t)r * d + cc];
    gv[i] = g[cc]; bv[i] = b[cc];
    biasv[i] = KS > 0 ? bias[cc] : 0.f;
#pragma unroll
    for (int s2 = 0; s2 < KS_MAX; s2++) t[i][s2] = (s2 < KS) ? P[(int64_t)s2 * plane + (int64_t)r * d + cc] : 0.f;
  }
  if (r >= n_live) return;
  float v[VPT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; i++) {
    const int c = tid + i * 256;
    v[i] = 0.f;
    if (c < d) {
      float a = xv[i];
      if (KS > 0) {                                  // x + (bias + partials), s ascending (mod.rs:346-348)
        float acc = biasv[i];
#pragma unroll
        for (int s2 = 0; s2 < KS_MAX; s2++) acc += t[i][s2];
        a = a + acc;
      }
      v[i] = a;
      s += a;
    }
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; i++) {
    const int c = tid + i * 256;
    if (c < d) { float a = v[i] - mean; q += a * a; }
  }
  q = wave_sum(q);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = q;
  __syncthreads();
  const float var = (red[4] + red[5] + red[6] + red[7]) / (float)d;
  const float denom = eps_inside_sqrt ? sqrtf(var + eps) : (sqrtf(var) + eps);
#pragma unroll
  for (int i = 0; i < VPT; i++) {
    const int c = tid + i * 256;
    if (c < d) {
      h[(int64_t)r * d + c] = (v[i] - mean) / denom * gv[i] + bv[i];
      x_out[(int64_t)r * d + c] = v[i];
    }
  }
}

// ---- skinny GEMM: P[ks][r][n] = sum_{k in slice ks} in[r][k] * W[k][n],  r < n_rows <= S -------
// block = 4 waves, column tile CT = 128 or 64 (LPR = CT / 4 lanes x float4 per weight row); a wave holds
// G = 64 / LPR lane groups, each owning 4 consecutive k per KH-deep step (KH = 16 G), two steps
// (8 x 16 B per lane) per round; input rows staged in LDS with the prologue applied.  LN prologue: one
// wave per row folds x + pending partials, takes the LayerNorm statistics with shuffles only, and stages
// the block's K-slice.  CT = 64 (logits) doubles the blocks per CU so that one block's prologue and
// statistics epilogue overlap another block's weight stream.
template <int MR, int XLD, int DPL, bool LN, bool STATS, int CT>
__global__ __launch_bounds__(256, (MR >= 8 && DPL >= 16) ? 1 : 2) void dec_gemv_kernel(GemvArgs a) {
  constexpr int LPR = CT / 4, G = 64 / LPR, KH = 16 * G;
  __shared__ __attribute__((aligned(16))) float xbuf[MR * XLD];   // input rows; later the cross-wave reduction buffer
  __shared__ float tilev[STATS ? MR : 1][CT];
  static_assert(MR * XLD >= 4 * MR * CT, "reduction buffer must fit");
  static_assert(CT == 128 || CT == 64, "column tile");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane / LPR, c4 = (lane % LPR) * 4;     // `half` = lane group inside the wave (0 .. G-1)
  const int ct = a.ct > 0 ? a.ct : CT;             // columns per block (<= CT)
  const int n0 = blockIdx.x * ct, ks = blockIdx.y;
  const int k0 = ks * a.KSL;
  const int kn = min(a.KSL, a.K - k0);          // rows of this slice (a multiple of KH)
  const int n_rows = a.st[ST_N];
  const int unit = wave * G + half;
  const bool col_ok = (n0 + c4) < a.ldw && c4 < ct;
  const int64_t ldw = a.ldw;
  // weight addressing: wave-uniform base (scalar registers) + one 32-bit per-lane offset, so a round's 8
  // loads share their address registers.  Lanes past the tile edge re-read the tile's first column
  // (always in range); their sums are discarded by the epilogue.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t voff = (uint32_t)(half * 4 * a.ldw + (col_ok ? c4 : 0));
  const int64_t wbase = (int64_t)(k0 + wave_u * G * 4) * ldw + n0;

  // weight tiles: NBUF register sets of 8 x 16 B per lane, software-pipelined -- set b is refilled for
  // round it + NBUF as soon as round it has consumed it, so (NBUF - 1) .. NBUF x 32 KB per block stay in flight
  constexpr int NBUF = STATS ? 3 : 2;
  float4 w[NBUF][8];
  auto load_round = [&](float4 (&wr)[8], int it) {   // rows k0 + unit * 4 + 2 KH it + {0..3, KH..KH+3}
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int ku = 2 * KH * it + (j >> 2) * KH;      // block-uniform; kn is a multiple of KH
      if (ku < kn) {
        const int64_t off = wbase + (int64_t)(ku + (j & 3)) * ldw;
        wr[j] = *reinterpret_cast<const float4*>(a.W + off + voff);
      }
    }
  };
  const int nit = (kn + 2 * KH - 1) / (2 * KH);   // rounds of 2 KH rows
  load_round(w[0], 0);   // first weight tiles in flight before the prologue touches memory
  if (n_rows == 0) return;   // chained decode whose windows have all finished (checked AFTER the loads were issued: the
                             // scalar state load must not sit in front of the weight stream)

  {
    // the host picks MR >= the live row count (one pass over the rows), or -- 9 - 16 live rows on the fused sublayer path --
    // row groups of MR in grid.z: group g stages and multiplies rows [g MR, g MR + MR)
    const int r0 = blockIdx.z * MR;
    if (r0 >= n_rows) return;
    // ---- prologue: stage in[0..MR)[k0..k0+kn) ----
    if constexpr (LN) {
      const int d = a.K;      // a multiple of 64 (session_reserve): lane + 64 i < d is wave-uniform
      const bool writer = (blockIdx.x == 0 && blockIdx.y == 0);
#pragma unroll 1
      for (int r = wave; r < MR; r += 4) {
        const int row = r0 + r;
        if (row >= n_rows) {   // padding row (wave-uniform)
          for (int c = lane; c < kn; c += 64) xbuf[r * XLD + c] = 0.f;
          continue;
        }
        // every load below is unconditional (columns past d alias column `lane`, planes past KSp alias the
        // last plane; both are masked after the load), so the whole prologue is one batch of loads in flight
        int co[DPL];
#pragma unroll
        for (int i = 0; i < DPL; i++) co[i] = lane + (64 * i < d ? 64 * i : 0);
        float gv[DPL], bv[DPL], v[DPL];
        const float* xr = a.src + (int64_t)row * d;
#pragma unroll
        for (int i = 0; i < DPL; i++) { gv[i] = a.ln_g[co[i]]; bv[i] = a.ln_b[co[i]]; v[i] = xr[co[i]]; }
        if (a.KSp > 0) {       // x + (attn/mlp output) = x + bias + sum_s partial_s, s ascending  (mod.rs:346-348)
          float acc[DPL];
#pragma unroll
          for (int i = 0; i < DPL; i++) acc[i] = a.pbias[co[i]];
          const float* pp = a.pend + (int64_t)row * d;
          const int64_t plane = (int64_t)a.S * d;
          // all loads of a chunk of partial planes are issued before the first add (memory-level
          // parallelism is what bounds this prologue); the summation order stays s ascending
          constexpr int CH = DPL <= 6 ? 8 : DPL <= 8 ? 4 : 2;   // <= 48 registers of partials in flight
          for (int sp = 0; sp < a.KSp; sp += CH) {
            float t[CH][DPL];
#pragma unroll
            for (int j = 0; j < CH; j++) {
              const float* pj = pp + (int64_t)min(sp + j, a.KSp - 1) * plane;
#pragma unroll
              for (int i = 0; i < DPL; i++) t[j][i] = pj[co[i]];
            }
#pragma unroll
            for (int j = 0; j < CH; j++) {
              const bool live = sp + j < a.KSp;
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
          if (64 * i < d) { s += v[i]; if (writer) a.x_out[(int64_t)row * d + co[i]] = v[i]; }
        }
        const float mean = wave_sum(s) / (float)d;          // Burn nn::LayerNorm: biased variance, two passes
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < DPL; i++) {
          if (64 * i < d) { const float t = v[i] - mean; q += t * t; }
        }
        const float var = wave_sum(q) / (float)d;
        const float denom = a.ln_inside ? sqrtf(var + a.ln_eps) : (sqrtf(var) + a.ln_eps);
#pragma unroll
        for (int i = 0; i < DPL; i++) {
          const int c = lane + 64 * i;
          if (c >= k0 && c < k0 + kn) xbuf[r * XLD + (c - k0)] = (v[i] - mean) / denom * gv[i] + bv[i];
        }
      }
    } else {
      for (int e = tid; e < MR * kn; e += 256) {
        const int r = e / kn, kk = e - r * kn;
        const int row = r0 + r, k = k0 + kk;
        float v = 0.f;
        if (row < n_rows) {
          if (a.pro == PRO_PLAIN) {
            v = a.src[(int64_t)row * a.ld_src + k];
          } else if (a.pro == PRO_GELU) {           // GELU(lin1(x)), mod.rs:377-378, lin1 partials folded here
            v = gelu_erf(fold_partials(a.src, a.KSp, (int64_t)a.S * a.ld_src, (int64_t)row * a.ld_src + k, a.pbias[k]));
          } else {                                  // PRO_ATTN: combine the key-chunk partials of cross-attention
            const int hh = k >> 6, dh = k & 63;
            v = combine_chunks(a.src + ((int64_t)(row * a.n_head + hh) * a.n_chunks) * CA_STRIDE, a.n_chunks, dh);
          }
        }
        xbuf[r * XLD + kk] = v;
      }
    }
    // the remaining register sets fill while the staged rows settle (they are dead during the prologue,
    // so the prologue's registers are reused)
#pragma unroll
    for (int b = 1; b < NBUF; b++)
      if (b < nit) load_round(w[b], b);
    __syncthreads();
    // ---- main loop ----
    float acc[MR][4];
#pragma unroll
    for (int r = 0; r < MR; r++) { acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.f; }
#pragma unroll 1
    for (int it = 0; it < nit; it += NBUF) {
#pragma unroll
      for (int b = 0; b < NBUF; b++) {
        if (it + b < nit) {                      // block-uniform
          const int kb = unit * 4 + 2 * KH * (it + b);
#pragma unroll
          for (int hstep = 0; hstep < 2; hstep++) {
            if (kb + hstep * KH < kn) {
#pragma unroll
              for (int r = 0; r < MR; r++) {
                const float4 xv = *reinterpret_cast<const float4*>(&xbuf[r * XLD + kb + hstep * KH]);
                const float4 wa = w[b][hstep * 4 + 0], wb_ = w[b][hstep * 4 + 1], wc = w[b][hstep * 4 + 2], wd = w[b][hstep * 4 + 3];
                acc[r][0] += xv.x * wa.x; acc[r][1] += xv.x * wa.y; acc[r][2] += xv.x * wa.z; acc[r][3] += xv.x * wa.w;
                acc[r][0] += xv.y * wb_.x; acc[r][1] += xv.y * wb_.y; acc[r][2] += xv.y * wb_.z; acc[r][3] += xv.y * wb_.w;
                acc[r][0] += xv.z * wc.x; acc[r][1] += xv.z * wc.y; acc[r][2] += xv.z * wc.z; acc[r][3] += xv.z * wc.w;
                acc[r][0] += xv.w * wd.x; acc[r][1] += xv.w * wd.y; acc[r][2] += xv.w * wd.z; acc[r][3] += xv.w * wd.w;
              }
            }
          }
          if (it + b + NBUF < nit) load_round(w[b], it + b + NBUF);
        }
      }
    }
    __syncthreads();   // everyone is done reading the input rows: reuse the buffer as red[4][MR][CT]
#pragma unroll
    for (int r = 0; r < MR; r++)
#pragma unroll
      for (int c = 0; c < 4; c++) {
        if constexpr (G == 4) acc[r][c] = xor16_sum(acc[r][c]);
        acc[r][c] = xor32_sum(acc[r][c]);
      }
    if (half == 0) {
#pragma unroll
      for (int r = 0; r < MR; r++)
        *reinterpret_cast<float4*>(&xbuf[(wave * MR + r) * CT + c4]) =
            make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
    }
    __syncthreads();
    for (int e = tid; e < MR * CT; e += 256) {
      const int r = e / CT, c = e - r * CT;
      const int row = r0 + r, col = n0 + c;
      float v = -INFINITY;
      if (row < n_rows && col < a.N && c < ct) {
        v = (xbuf[(0 * MR + r) * CT + c] + xbuf[(1 * MR + r) * CT + c]) +
            (xbuf[(2 * MR + r) * CT + c] + xbuf[(3 * MR + r) * CT + c]);
        a.P[((int64_t)ks * a.S + row) * a.N + col] = v;
        if constexpr (STATS) { if (a.use_mask) v += a.mask[col]; }   // transcribe.rs:271-275
      }
      if constexpr (STATS) tilev[r][c] = v;
    }
    __syncthreads();
    if constexpr (STATS) {
      // per-tile log-softmax statistics and top-k candidates of each row (one wave per row)
      for (int r = wave; r < MR; r += 4) {
        const int row = r0 + r;
        if (row >= n_rows) continue;
        float v0 = tilev[r][lane], v1 = CT > 64 ? tilev[r][(lane + 64) % CT] : -INFINITY;
        const float m = wave_max(fmaxf(v0, v1));
        const float se = m > -INFINITY ? wave_sum(expf(v0 - m) + expf(v1 - m)) : 0.f;
        float* ts = a.tstats + ((int64_t)row * gridDim.x + blockIdx.x) * TS_STRIDE;
        if (lane == 0) { ts[0] = m; ts[1] = se; }
        for (int j = 0; j < a.topk; j++) {
          float bv; int bi;
          if (better(v0, lane, v1, lane + 64)) { bv = v0; bi = lane; } else { bv = v1; bi = lane + 64; }
          wave_argmax(bv, bi);
          if (lane == 0) { ts[2 + 2 * j] = bv; ts[3 + 2 * j] = __int_as_float(n0 + bi); }
          if (bi == lane) v0 = -INFINITY;
          if (bi == lane + 64) v1 = -INFINITY;
        }
      }
      __syncthreads();
    }
  }
}

// The final fold + LayerNorm of the 9 - 16-row logits pass, ONCE: one wave per row, the arithmetic of the logits prologue
// (planes in ascending order, statistics by wave shuffles) -> h_tmp [rows][d], and the folded stream to x_out.
template <int DPL>
__global__ __launch_bounds__(64) void dec_fold_ln_rows_kernel(GemvArgs a) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const int d = a.K;
  if (r >= a.st[ST_N]) return;
  int co[DPL];
#pragma unroll
  for (int i = 0; i < DPL; i++) co[i] = lane + (64 * i < d ? 64 * i : 0);
  float gv[DPL], bv[DPL], v[DPL];
  const float* xr = a.src + (int64_t)r * d;
#pragma unroll
  for (int i = 0; i < DPL; i++) { gv[i] = a.ln_g[co[i]]; bv[i] = a.ln_b[co[i]]; v[i] = xr[co[i]]; }
  if (a.KSp > 0) {
    float acc[DPL];
#pragma unroll
    for (int i = 0; i < DPL; i++) acc[i] = a.pbias[co[i]];
    const float* pp = a.pend + (int64_t)r * d;
    const int64_t plane = (int64_t)a.S * d;
    constexpr int CH = DPL <= 6 ? 8 : 4;
    for (int sp = 0; sp < a.KSp; sp += CH) {
      float t[CH][DPL];
#pragma unroll
      for (int j = 0; j < CH; j++) {
        const float* pj = pp + (int64_t)min(sp + j, a.KSp - 1) * plane;
#pragma unroll
        for (int i = 0; i < DPL; i++) t[j][i] = pj[co[i]];
      }
#pragma unroll
      for (int j = 0; j < CH; j++) {
        const bool live = sp + j < a.KSp;
#pragma unroll
        for (int i = 0; i < DPL; i++) acc[i] += live ? t[j][i] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < DPL; i++) v[i] = v[i] + acc[i];
  }
  float sm = 0.f;
#pragma unroll
  for (int i = 0; i < DPL; i++) {
    if (64 * i < d) { sm += v[i]; a.x_out[(int64_t)r * d + co[i]] = v[i]; }
  }
  const float mean = wave_sum(sm) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < DPL; i++) {
    if (64 * i < d) { const float t = v[i] - mean; q += t * t; }
  }
  const float var = wave_sum(q) / (float)d;
  const float denom = a.ln_inside ? sqrtf(var + a.ln_eps) : (sqrtf(var) + a.ln_eps);
#pragma unroll
  for (int i = 0; i < DPL; i++) {
    const int c = lane + 64 * i;
    if (c < d) a.h_tmp[(int64_t)r * d + c] = (v[i] - mean) / denom * gv[i] + bv[i];
  }
}

// ---- logits for 9 - 16 live rows on the matrix cores (beam search over a 30 s chunk: 3 windows x 5 beams = 15 rows) -------
// The vector-pipe GEMV above spends 16 x 128 x d FMAs per block with an LDS read per four of them: 41 us per step for
// tiny.en's 80 MB of E^T even without its prologue (profiles/r05_k_fold16.txt).  Same launch geometry (one block per
// 128-column tile), same prologue (fold of the pending planes + LayerNorm, one wave per row), same tile statistics and
// output planes -- but the product runs on v_mfma_f32_16x16x4_f32 (exact f32, the skinny weight-stream kernel's scheme,
// decode_batch.hip): a wave owns 64 columns over half of K, one float4 per lane = 4 K-rows x 64 columns per load, component c
// of the float4 is the B operand of accumulator c (columns 4 j + c), the 16 rows sit k-major in LDS (A operand of lane l for
// K-rows k .. k + 3 is word 16 k + l: conflict-free).  The two K halves of a column strip meet in LDS in fixed order.
typedef float lg_f32x4 __attribute__((ext_vector_type(4)));
// PRELN: the rows arrive folded and normalised (a.h_tmp, written by dec_fold_ln_rows_kernel one launch earlier) -- with the
// product on the matrix cores the per-block prologue was the larger half of the launch (every one of the ~400 blocks folding
// 16 rows x 24 planes from L2: 250 MB of traffic for an 80 MB weight stream; 41 us with it in, profiles/r06_c_bench.json).
template <int DPL, bool PRELN>
__global__ __launch_bounds__(256, 2) void dec_logits_mfma16_kernel(GemvArgs a) {
  constexpr int MR = 16, CT = 128, KMAX = 64 * DPL;
  __shared__ __attribute__((aligned(16))) float xT[KMAX * MR];          // [k][16]; later red[2][MR][CT] (KMAX >= 256)
  __shared__ float tilev[MR][CT];
  static_assert(KMAX * MR >= 2 * MR * CT, "reduction buffer must fit");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ct = a.ct > 0 ? a.ct : CT;
  const int n0 = blockIdx.x * ct;
  const int d = a.K;                                 // a multiple of 64; the host guarantees d % 8 == 0 and d <= KMAX
  const int n_rows = a.st[ST_N];
  const int strip = wave & 1, khalf = wave >> 1;
  const int kh = d >> 1;                             // K-rows per wave (a multiple of 4: d % 8 == 0)
  const int krow = lane >> 4, cq = lane & 15;
  const int col0 = strip * 64 + 4 * cq;              // first of this lane's four columns inside the tile
  const bool col_ok = (n0 + col0) < a.ldw && col0 < ct;
  const float* bp = a.W + (int64_t)(khalf * kh + krow) * a.ldw + n0 + (col_ok ? col0 : 0);
  // float4 loads a lane keeps in flight per pass.  Measured at tiny.en (profiles/r06_r_bench_logits_ld48.json,
  // r06_s_ab_logits_ld.txt): the wave's whole K half in one pass (48 loads, 234 VGPRs, two waves per SIMD) 35.8 us; two passes
  // of 24 (168 VGPRs) 31.0; three of 16 30.0; four of 12 29.5 -- occupancy is worth more than the extra round trips
#ifndef WB_LOGITS_LD
#define WB_LOGITS_LD 12
#endif
  constexpr int LD = WB_LOGITS_LD;
  const int nld = kh >> 2;
  float4 bw[LD];
#pragma unroll
  for (int t = 0; t < LD; t++)
    if (t < nld) bw[t] = *reinterpret_cast<const float4*>(bp + (int64_t)(4 * t) * a.ldw);
  if (n_rows == 0) return;
  // ---- prologue: x + (bias + pending planes), LayerNorm, staged k-major (dec_gemv_kernel's LN prologue) ----
  if constexpr (PRELN) {
    for (int e = tid; e < MR * d; e += 256) {
      const int r = e / d, c = e - r * d;
      xT[c * MR + r] = r < n_rows ? a.h_tmp[(int64_t)r * d + c] : 0.f;
    }
  } else {
    const bool writer = blockIdx.x == 0;
#pragma unroll 1
    for (int r = wave; r < MR; r += 4) {
      if (r >= n_rows) {
        for (int c = lane; c < d; c += 64) xT[c * MR + r] = 0.f;
        continue;
      }
      int co[DPL];
#pragma unroll
      for (int i = 0; i < DPL; i++) co[i] = lane + (64 * i < d ? 64 * i : 0);
      float gv[DPL], bv[DPL], v[DPL];
      const float* xr = a.src + (int64_t)r * d;
#pragma unroll
      for (int i = 0; i < DPL; i++) { gv[i] = a.ln_g[co[i]]; bv[i] = a.ln_b[co[i]]; v[i] = xr[co[i]]; }
      if (a.KSp > 0) {
        float acc[DPL];
#pragma unroll
        for (int i = 0; i < DPL; i++) acc[i] = a.pbias[co[i]];
        const float* pp = a.pend + (int64_t)r * d;
        const int64_t plane = (int64_t)a.S * d;
        constexpr int CH = DPL <= 6 ? 8 : 4;
        for (int sp = 0; sp < a.KSp; sp += CH) {
          float t[CH][DPL];
#pragma unroll
          for (int j = 0; j < CH; j++) {
            const float* pj = pp + (int64_t)min(sp + j, a.KSp - 1) * plane;
#pragma unroll
            for (int i = 0; i < DPL; i++) t[j][i] = pj[co[i]];
          }
#pragma unroll
          for (int j = 0; j < CH; j++) {
            const bool live = sp + j < a.KSp;
#pragma unroll
            for (int i = 0; i < DPL; i++) acc[i] += live ? t[j][i] : 0.f;
          }
        }
#pragma unroll
        for (int i = 0; i < DPL; i++) v[i] = v[i] + acc[i];
      }
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < DPL; i++) {
        if (64 * i < d) { sm += v[i]; if (writer) a.x_out[(int64_t)r * d + co[i]] = v[i]; }
      }
      const float mean = wave_sum(sm) / (float)d;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < DPL; i++) {
        if (64 * i < d) { const float t = v[i] - mean; q += t * t; }
      }
      const float var = wave_sum(q) / (float)d;
      const float denom = a.ln_inside ? sqrtf(var + a.ln_eps) : (sqrtf(var) + a.ln_eps);
#pragma unroll
      for (int i = 0; i < DPL; i++) {
        const int c = lane + 64 * i;
        if (c < d) xT[c * MR + r] = (v[i] - mean) / denom * gv[i] + bv[i];
      }
    }
  }
  __syncthreads();
  // ---- product: 4 accumulators (columns 4 j + c of the strip), K-rows in ascending order ----
  lg_f32x4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; c++) acc[c] = lg_f32x4{0.f, 0.f, 0.f, 0.f};
  const float* ap = xT + (khalf * kh) * MR + lane;
  // (pass-synchronous: a rolling ring -- slot t refilled right behind its use -- measured 36.7 against 31.0 us per launch,
  // profiles/r06_j_bench.json)
  for (int t0 = 0; t0 < nld; t0 += LD) {
#pragma unroll
    for (int t = 0; t < LD; t++) {
      if (t0 + t < nld) {
        const float av = ap[(4 * (t0 + t)) * MR];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw[t].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw[t].y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw[t].z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw[t].w, acc[3], 0, 0, 0);
      }
    }
    if (t0 + LD < nld) {
#pragma unroll
      for (int t = 0; t < LD; t++)
        if (t0 + LD + t < nld) bw[t] = *reinterpret_cast<const float4*>(bp + (int64_t)(4 * (t0 + LD + t)) * a.ldw);
    }
  }
  __syncthreads();                                   // everyone is done with the staged rows: reuse them as red[2][MR][CT]
  // accumulator c, register i of lane l: row 4 (l / 16) + i, column 4 (l % 16) + c of the strip
#pragma unroll
  for (int i = 0; i < 4; i++)
    *reinterpret_cast<float4*>(&xT[(khalf * MR + 4 * krow + i) * CT + col0]) = make_float4(acc[0][i], acc[1][i], acc[2][i], acc[3][i]);
  __syncthreads();
  for (int e = tid; e < MR * CT; e += 256) {
    const int r = e / CT, c = e - r * CT;
    const int col = n0 + c;
    float v = -INFINITY;
    if (r < n_rows && col < a.N && c < ct) {
      v = xT[(0 * MR + r) * CT + c] + xT[(1 * MR + r) * CT + c];
      a.P[(int64_t)r * a.N + col] = v;
      if (a.use_mask) v += a.mask[col];              // transcribe.rs:271-275
    }
    tilev[r][c] = v;
  }
  __syncthreads();
  // per-tile log-softmax statistics and top-k candidates of each row (one wave per row; dec_gemv_kernel's epilogue)
  for (int r = wave; r < MR; r += 4) {
    if (r >= n_rows) continue;
    float v0 = tilev[r][lane], v1 = tilev[r][lane + 64];
    const float m = wave_max(fmaxf(v0, v1));
    const float se = m > -INFINITY ? wave_sum(expf(v0 - m) + expf(v1 - m)) : 0.f;
    float* ts = a.tstats + ((int64_t)r * gridDim.x + blockIdx.x) * TS_STRIDE;
    if (lane == 0) { ts[0] = m; ts[1] = se; }
    for (int j = 0; j < a.topk; j++) {
      float bv; int bi;
      if (better(v0, lane, v1, lane + 64)) { bv = v0; bi = lane; } else { bv = v1; bi = lane + 64; }
      wave_argmax(bv, bi);
      if (lane == 0) { ts[2 + 2 * j] = bv; ts[3 + 2 * j] = __int_as_float(n0 + bi); }
      if (bi == lane) v0 = -INFINITY;
      if (bi == lane + 64) v1 = -INFINITY;
    }
  }
}

// device-chained greedy: the argmax feeds the next step, tokens stay on the device
// When the last unfinished window ends, the step state is blanked (ST_N = 0): the host enqueues decode chunks ahead
// of reading the finished flags, and every kernel of an already-enqueued step then exits at its first instruction.
// `len`, `finished` and `step` are the row's state words as read at kernel entry (before any block of this launch has
// written state): the update itself is stores only -- three dependent global reads by one thread used to sit between the
// top-1 and the next step's embedding gather.
// A top-1 without a finite candidate keeps the candidate lists' initial id 0x7fffffff: the row's logits were NaN (an
// activation left the range of a 16-bit GEMM path and the range guard is about to fail the call, or the weights are not
// finite).  That id must never become an embedding index (dec_prepare_kernel / the merge kernel's next-step gather read
// E + tok * d): the row ends on <|endoftext|> instead and GC_BAD makes the host fail the call loudly (session_greedy_chain).
__device__ __forceinline__ int top1_or_eot(int gi, int eot, int* gctl) {
  if ((unsigned)gi >= 0x7fffffffu) { gctl[GC_BAD] = 1; return eot; }
  return gi;
}

__device__ __forceinline__ int chained_update(int* st, const StepLayout& lay, int* gctl, int* gtok, int Lmax,
                                              int eot, int r, int gi, int len, int finished, int step) {
  if (!finished) {
    // (a finished row keeps its last token: its attention blocks no longer run, so its later "argmax" is computed
    // from stale planes and must never become an embedding index)
    gctl[GC_HDR + r] = gi;
    gtok[r * Lmax + len] = gi;
    gctl[GC_HDR + 2 * lay.S + r] = len + 1;
    if (gi == eot) {
      finished = 1;
      gctl[GC_HDR + lay.S + r] = 1;                // finished (transcribe.rs:235-241): later tokens are ignored
      if (atomicAdd(&gctl[GC_NDONE], 1) + 1 == lay.W) { gctl[GC_ALLDONE] = 1; st[ST_N] = 0; }
    }
  }
  if (r == 0) gctl[GC_STEP] = step + 1;
  return finished;
}
// host-visible progress of row r (mapped pinned memory): finished flag first, then the step counter the host waits for
__device__ __forceinline__ void chained_publish(int* hflags, int r, int finished, int step) {
  __hip_atomic_store(&hflags[2 * r + 1], finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&hflags[2 * r], step + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- merge the per-tile statistics of one beam's logits row: log_softmax + top-k ------------------
// Device-chained greedy decode (gctl != nullptr) with `nx.x` set: the block of row r also PREPARES row r of the
// next step (token + position embedding, position table, step state), which saves the prepare launch.
// Cross-block hazards: every block only reads and writes its own row's state; ST_STEP is read and written by block 0
// only.  ST_N is written by ONE block at most once per chain -- blanked to 0 by the block whose row is the LAST window
// to finish (GC_NDONE == W) -- while sibling blocks of the same launch may still be reading it at kernel entry.  That is
// benign by this invariant: a block that reads 0 instead of n belongs to a row that had ALREADY finished in an earlier
// step (the last finisher itself read n), so the only thing it skips is a chained_update that would have been a no-op
// and a host-flag publish the host does not wait for (wait_flags ends through its "every window finished" branch).
__global__ __launch_bounds__(256) void dec_topk_merge_kernel(int* __restrict__ st, const float* __restrict__ tstats,
                                                             int n_tiles, int k, int32_t* __restrict__ out_id,
                                                             float* __restrict__ out_lp, float* __restrict__ row_stats,
                                                             StepLayout lay, int* __restrict__ gctl,
                                                             int* __restrict__ gtok, int Lmax, int eot, NextPrep nx) {
  __shared__ float redv[4];
  __shared__ int redi[4];
  __shared__ float bc[2];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_live = st[ST_N], len_now = st[lay.len + r];      // requested together with the tile records below
  const int step_now = st[ST_STEP];
  const int fin_now = gctl ? gctl[GC_HDR + lay.S + r] : 0;
  const int st_tok_now = st[lay.tok + r];
  const float* ts = tstats + (int64_t)r * n_tiles * TS_STRIDE;
  // Ordinary vocabularies give <= 512 tiles: each thread pulls the whole record of its (at most) two tiles -- max,
  // sum-exp and the k candidates -- in ONE batch of loads, so the row costs one global round trip instead of three
  // dependent passes over the same records.
  constexpr int TPT = 2;
  const bool fast = n_tiles <= 256 * TPT;
  float pm[TPT], psum[TPT], pcv[TPT][TOPK_MAX];
  int pci[TPT][TOPK_MAX];
  if (fast) {
#pragma unroll
    for (int j = 0; j < TPT; j++) {
      const int t = tid + 256 * j;
      const bool ok = t < n_tiles;
      const float* rec = ts + (int64_t)(ok ? t : 0) * TS_STRIDE;
      pm[j] = ok ? rec[0] : -INFINITY;
      psum[j] = ok ? rec[1] : 0.f;
#pragma unroll
      for (int q = 0; q < TOPK_MAX; q++) {
        const bool okq = ok && q < k;
        pcv[j][q] = okq ? rec[2 + 2 * q] : -INFINITY;
        pci[j][q] = okq ? __float_as_int(rec[3 + 2 * q]) : 0x7fffffff;
      }
    }
  }
  if (r >= n_live) return;
  float m = -INFINITY;
  if (fast) {
#pragma unroll
    for (int j = 0; j < TPT; j++) m = fmaxf(m, pm[j]);
  } else {
    for (int t = tid; t < n_tiles; t += 256) m = fmaxf(m, ts[t * TS_STRIDE]);
  }
  m = wave_max(m);
  if (lane == 0) redv[wave] = m;
  __syncthreads();
  const float M = fmaxf(fmaxf(redv[0], redv[1]), fmaxf(redv[2], redv[3]));
  float s = 0.f;
  if (fast) {
#pragma unroll
    for (int j = 0; j < TPT; j++)
      if (pm[j] > -INFINITY) s += expf(pm[j] - M) * psum[j];
  } else {
    for (int t = tid; t < n_tiles; t += 256) {
      const float mt = ts[t * TS_STRIDE];
      if (mt > -INFINITY) s += expf(mt - M) * ts[t * TS_STRIDE + 1];
    }
  }
  s = wave_sum(s);
  __syncthreads();
  if (lane == 0) redv[wave] = s;
  __syncthreads();
  if (tid == 0) {
    bc[0] = logf((redv[0] + redv[1]) + (redv[2] + redv[3]));
    row_stats[2 * r] = M; row_stats[2 * r + 1] = bc[0];
  }
  // local top-k over this thread's candidates (value desc, id asc): a strict total order, so the merged result does
  // not depend on how the candidates are dealt to the threads
  float tv[TOPK_MAX]; int ti[TOPK_MAX];
#pragma unroll
  for (int j = 0; j < TOPK_MAX; j++) { tv[j] = -INFINITY; ti[j] = 0x7fffffff; }
  auto insert = [&](float cv, int ci) {
    if (cv > -INFINITY) {
#pragma unroll
      for (int q = 0; q < TOPK_MAX; q++)
        if (better(cv, ci, tv[q], ti[q])) { float a = tv[q]; int b = ti[q]; tv[q] = cv; ti[q] = ci; cv = a; ci = b; }
    }
  };
  if (fast) {
#pragma unroll
    for (int j = 0; j < TPT; j++)
#pragma unroll
      for (int q = 0; q < TOPK_MAX; q++) insert(pcv[j][q], pci[j][q]);
  } else {
    for (int c = tid; c < n_tiles * k; c += 256) {
      const int t = c / k, j = c - t * k;
      insert(ts[t * TS_STRIDE + 2 + 2 * j], __float_as_int(ts[t * TS_STRIDE + 3 + 2 * j]));
    }
  }
  __syncthreads();
  const float lse = bc[0];
  int first = 0;
  for (int round = 0; round < k; round++) {
    float bv = tv[0]; int bi = ti[0];
    wave_argmax(bv, bi);
    __syncthreads();
    if (lane == 0) { redv[wave] = bv; redi[wave] = bi; }
    __syncthreads();
    float gv = redv[0]; int gi = redi[0];
    for (int j = 1; j < 4; j++)
      if (better(redv[j], redi[j], gv, gi)) { gv = redv[j]; gi = redi[j]; }
    const int gi_list = gi;                    // (what the candidate lists hold: the pop below compares with it)
    if (gctl && round == 0) gi = top1_or_eot(gi, eot, gctl);
    if (tid == 0) {
      out_id[r * TOPK_MAX + round] = gi;
      out_lp[r * TOPK_MAX + round] = (gv - M) - lse;   // log_softmax, transcribe.rs:276
      if (gctl && round == 0) {
        const int fin = chained_update(st, lay, gctl, gtok, Lmax, eot, r, gi, len_now, fin_now, step_now);
        if (nx.hflags) chained_publish(nx.hflags, r, fin, step_now);
      }
    }
    if (round == 0) first = gi;
    if (ti[0] == gi_list) {   // the winner pops its head
#pragma unroll
      for (int j = 0; j < TOPK_MAX - 1; j++) { tv[j] = tv[j + 1]; ti[j] = ti[j + 1]; }
      tv[TOPK_MAX - 1] = -INFINITY; ti[TOPK_MAX - 1] = 0x7fffffff;
    }
  }
  if (gctl && nx.x && len_now < Lmax) {
    // next step of the chain: position len_now (0-based) holds token `first`
    __syncthreads();                       // chained_update (thread 0) has read this row's state
    const int step = len_now - 1, nstep = step + 1;
    const int tok_next_w = fin_now ? st_tok_now : first;
    if (tid == 0) {
      st[lay.tok + r] = tok_next_w;
      st[lay.len + r] = len_now + 1;
      st[lay.dead + r] = (fin_now || first == eot) ? 1 : 0;          // = chained_update's `finished` for this row
      if (r == 0) st[ST_STEP] = nstep;
    }
    int* tab_new = nx.tabs + (size_t)(nstep & 1) * lay.S * Lmax;
    const int* tab_old = nx.tabs + (size_t)((nstep & 1) ^ 1) * lay.S * Lmax;
    for (int p = tid; p < len_now; p += 256) tab_new[r * Lmax + p] = tab_old[r * Lmax + p];
    if (tid == 0) tab_new[r * Lmax + len_now] = nstep * lay.S + r;
    const float4* e = reinterpret_cast<const float4*>(nx.E + (int64_t)tok_next_w * nx.d);
    const float4* pp = reinterpret_cast<const float4*>(nx.pos + (int64_t)len_now * nx.d);
    float4* o = reinterpret_cast<float4*>(nx.x + (int64_t)r * nx.d);
    for (int c = tid; c < (nx.d >> 2); c += 256) {
      float4 a = e[c], b = pp[c];
      o[c] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
  }
}

// ---- masked self-attention over the paged self-KV cache, one block per (beam, head) --------------
constexpr int SA_MAXPOS = 448;
__global__ __launch_bounds__(256) void dec_self_attn_kernel(const int* __restrict__ st, StepLayout lay,
                                                            const float* __restrict__ Pqkv, int KS,
                                                            const float* __restrict__ bqkv, int d,
                                                            float* __restrict__ Kc, float* __restrict__ Vc,
                                                            const int* __restrict__ tabs, int Lmax, float scale,
                                                            float* __restrict__ att) {
  __shared__ __attribute__((aligned(16))) float qkv[3][64];   // q*s, k*s, v of the new token
  __shared__ int tbs[SA_MAXPOS];
  __shared__ float ps[SA_MAXPOS];
  __shared__ float red[8];
  __shared__ float ored[4][64];
  const int i = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (i >= st[ST_N] || st[lay.dead + i]) return;   // (dead: a chained row whose window has ended)
  const int len = st[lay.len + i];
  const int* tb = tabs + (size_t)(st[ST_STEP] & 1) * lay.S * Lmax + i * Lmax;
  for (int p = tid; p < len; p += 256) tbs[p] = tb[p];
  if (tid < 192) {   // fold the QKV partials: q = (xWq + bq) * s, k = (xWk) * s, v = xWv + bv  (mod.rs:429-431, :506-514)
    const int which = tid >> 6, col = which * d + h * 64 + lane;
    float v = fold_partials(Pqkv, KS, (int64_t)lay.S * 3 * d, (int64_t)i * 3 * d + col, bqkv[col]);
    if (which < 2) v *= scale;
    qkv[which][lane] = v;
  }
  __syncthreads();
  if (tid >= 64 && tid < 192) {   // append to the cache
    const int which = tid >> 6;
    float* dst = (which == 1 ? Kc : Vc) + (int64_t)tbs[len - 1] * d + h * 64 + lane;
    *dst = qkv[which][lane];
  }
  // this thread's V column for positions p = wave (mod 4): requested BEFORE the K rows are consumed, so the two
  // cache streams share one round trip (they only depend on the position table)
  const int col = h * 64 + lane;
  float vpre[28];
#pragma unroll
  for (int i = 0; i < 28; i++) {
    const int p = wave + 4 * i;
    vpre[i] = p < len - 1 ? Vc[(int64_t)tbs[p] * d + col] : 0.f;
  }
  // scores: one position per thread (len <= 448 -> at most 2 passes)
  float sc[2];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int p = tid + j * 256;
    sc[j] = -INFINITY;
    if (p < len) {
      float acc = 0.f;
      if (p == len - 1) {
#pragma unroll
        for (int c = 0; c < 64; c++) acc += qkv[0][c] * qkv[1][c];
      } else {
        const float4* kr = reinterpret_cast<const float4*>(Kc + (int64_t)tbs[p] * d + h * 64);
        float4 kv[16];
#pragma unroll
        for (int c = 0; c < 16; c++) kv[c] = kr[c];
#pragma unroll
        for (int c = 0; c < 16; c++) {
          const float4 qv = *reinterpret_cast<const float4*>(&qkv[0][4 * c]);
          acc += qv.x * kv[c].x + qv.y * kv[c].y + qv.z * kv[c].z + qv.w * kv[c].w;
        }
      }
      sc[j] = acc;
      m = fmaxf(m, acc);
    }
  }
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int p = tid + j * 256;
    if (p < len) { const float e = expf(sc[j] - m); ps[p] = e; l += e; }
  }
  l = wave_sum(l);
  if (lane == 0) red[4 + wave] = l;
  __syncthreads();
  l = (red[4] + red[5]) + (red[6] + red[7]);
  // o[dh] = sum_p ps[p] V[p][dh]: wave g takes positions p = g (mod 4), four loads in flight
  float o = 0.f;
#pragma unroll
  for (int i = 0; i < 28; i++) {
    const int p = wave + 4 * i;
    if (p < len - 1) o += ps[p] * vpre[i];
  }
  for (int p = wave + 4 * 28; p < len - 1; p += 4) o += ps[p] * Vc[(int64_t)tbs[p] * d + col];   // len > 112
  if (wave == ((len - 1) & 3)) o += ps[len - 1] * qkv[2][lane];
  ored[wave][lane] = o;
  __syncthreads();
  if (tid < 64) att[(int64_t)i * d + col] = ((ored[0][lane] + ored[1][lane]) + (ored[2][lane] + ored[3][lane])) / l;
}

// ---- cross-attention over one key chunk of one window's cached K/V, all of its beams -----
constexpr int CA_CH = 128;   // keys per chunk
// KQ > 0 (= d / 4, small models): the query projection is FUSED -- the block folds the residual stream and the
// pending out-projection partials of its window's beams, applies cross_attn_ln, and multiplies by the 64 columns
// of Wq that belong to its head (thread = output column x quarter of K, the weight slice prefetched into
// registers at kernel start); block (chunk 0, head 0) writes the folded residual stream.  This saves the
// LN + Wq GEMV launch of every decoder layer; the slice is re-read by the key-chunk blocks of the head (L2).
// fz.Wo != nullptr: the block also applies its head's 64 rows of the out-projection to its (unnormalised) chunk output
// and writes {m, l, o_chunk Wo[head rows, :]} records -- the out-projection GEMV launch disappears, the consumer
// (dec_mlp_fused_kernel) combines the chunk records of all heads while it folds the residual stream.
template <int NB, int KQ>    // NB: register-resident beams per window (>= the largest live count this step)
__global__ __launch_bounds__(256) void dec_cross_attn_kernel(const int* __restrict__ st, StepLayout lay,
                                                             const float* __restrict__ Pq, int KS,
                                                             const float* __restrict__ bq, int d,
                                                             const float* __restrict__ ckv, int ldkv, int koff,
                                                             const int* __restrict__ win_row0,
                                                             const int* __restrict__ win_C, float scale,
                                                             int n_head, int n_chunks, float* __restrict__ ca,
                                                             CaFuse fz) {
  // per-beam arrays are sized by NB, not MAX_BEAMS: the one-beam-per-window instance (greedy batch mode: 38-51 windows
  // x heads x chunks = thousands of blocks streaming the cached K/V) then needs 36 KB instead of 55 KB of LDS and
  // four blocks share a CU instead of two -- twice the loads in flight while a block sits in its softmax phases
  __shared__ __attribute__((aligned(16))) float hs[KQ > 0 ? NB : 1][KQ > 0 ? 4 * KQ : 1];   // cross_attn_ln(x) rows
  __shared__ __attribute__((aligned(16))) float Kt[CA_CH][65];
  __shared__ __attribute__((aligned(16))) float qs[NB][64];
  __shared__ float sp[2][NB][CA_CH];
  __shared__ float pb[NB][CA_CH];
  __shared__ float stat[NB][2];
  __shared__ __attribute__((aligned(16))) float ored[4][NB][64];
  const int c = blockIdx.x, h = blockIdx.y, w = blockIdx.z, tid = threadIdx.x;
  const int nb = st[lay.win_nb + w];
  if (nb == 0 || st[ST_N] == 0) return;            // (ST_N == 0: a chained decode whose windows have all finished)
  const int* slots = st + lay.win_slots + w * MAX_BEAMS;
  const int C = win_C[w];
  const int j0 = c * CA_CH;
  const int nk = min(CA_CH, C - j0);
  if (nk <= 0) {   // chunk past this window's encoder length: neutral partial
    if (fz.Wo != nullptr) {
      const int64_t rstride = d + 2;
      for (int e = tid; e < nb * (int)rstride; e += 256) {
        const int b = e / (int)rstride, f = e - b * (int)rstride;
        fz.rec[((int64_t)(h * n_chunks + c) * lay.S + slots[b]) * rstride + f] = (f == 0) ? -1.0e30f : 0.f;
      }
      return;
    }
    for (int e = tid; e < nb * CA_STRIDE; e += 256) {
      const int b = e / CA_STRIDE, f = e - b * CA_STRIDE;
      ca[((int64_t)(slots[b] * n_head + h) * n_chunks + c) * CA_STRIDE + f] = (f == 0) ? -1.0e30f : 0.f;
    }
    return;
  }
  const float* Kb = ckv + (int64_t)(win_row0[w] + j0) * ldkv + koff + h * 64;   // K pre-scaled at projection time
  const float* Vb = Kb + d;
  // issue the K tile and this thread's V column first; the q fold rides under their latency
  float4 kreg[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int e = tid + i * 256, r = e >> 4, q4 = (e & 15) * 4;
    kreg[i] = r < nk ? *reinterpret_cast<const float4*>(Kb + (int64_t)r * ldkv + q4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int dh_t = tid & 63, gq = tid >> 6;
  float vreg[32];
#pragma unroll
  for (int i = 0; i < 32; i++) {
    const int j = gq * 32 + i;
    vreg[i] = j < nk ? Vb[(int64_t)j * ldkv + dh_t] : 0.f;
  }
  // fused out-projection (KQ > 0 variants): thread (cf, jg) owns a float4 of output columns and RPGW rows of Wo's
  // head slice; requested as soon as the Wq registers are free, consumed in the epilogue
  constexpr int DD = KQ > 0 ? 4 * KQ : 128, CFW = DD / 4, GW = 256 / CFW >= 8 ? 8 : 2, RPGW = 64 / GW;
  const int cfw = tid % CFW, jgw = tid / CFW;
  const bool actw = jgw < GW;
  float4 wo[KQ > 0 ? RPGW : 1];
  if constexpr (KQ > 0) {
    // this thread's slice of Wq: column h*64 + dh_t, rows gq*KQ .. +KQ (in flight with everything above; 16-byte
    // loads with 16 K-groups per block measured slower: 1167x vs 1308x)
    float wq[KQ];
    const float* wcol = fz.Wq + (int64_t)(gq * KQ) * d + h * 64 + dh_t;
#pragma unroll
    for (int i = 0; i < KQ; i++) wq[i] = wcol[(int64_t)i * d];
    // x + (out-projection of self-attention) -> cross_attn_ln  (mod.rs:346-348), one wave per beam
    {
      constexpr int DPL = KQ / 16;                    // d / 64 columns per lane
      const int wave = tid >> 6, lane = tid & 63;
      for (int b = wave; b < nb; b += 4) {
        const int row = slots[b];
        float v[DPL];
#pragma unroll
        for (int i = 0; i < DPL; i++) {
          const int col = lane + 64 * i;
          v[i] = fz.x_in[(int64_t)row * d + col] +
                 fold_partials(fz.pend, fz.KSp, (int64_t)lay.S * d, (int64_t)row * d + col, fz.pbias[col]);
        }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < DPL; i++) {
          sum += v[i];
          if (c == 0 && h == 0) fz.x_out[(int64_t)row * d + lane + 64 * i] = v[i];
        }
        const float mean = wave_sum(sum) / (float)d;
        float q2 = 0.f;
#pragma unroll
        for (int i = 0; i < DPL; i++) { const float t = v[i] - mean; q2 += t * t; }
        const float var = wave_sum(q2) / (float)d;
        const float denom = fz.ln_inside ? sqrtf(var + fz.ln_eps) : (sqrtf(var) + fz.ln_eps);
#pragma unroll
        for (int i = 0; i < DPL; i++) {
          const int col = lane + 64 * i;
          hs[b][col] = (v[i] - mean) / denom * fz.ln_g[col] + fz.ln_b[col];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int e = tid + i * 256, r = e >> 4, q4 = (e & 15) * 4;
      Kt[r][q4 + 0] = kreg[i].x; Kt[r][q4 + 1] = kreg[i].y; Kt[r][q4 + 2] = kreg[i].z; Kt[r][q4 + 3] = kreg[i].w;
    }
    __syncthreads();
    {   // q = (cross_attn_ln(x) Wq + bq) * s  (mod.rs:483, :506-509): quarter-K partial sums, then the four quarters
      float acc[NB];
#pragma unroll
      for (int b = 0; b < NB; b++) acc[b] = 0.f;
#pragma unroll
      for (int i = 0; i < KQ; i += 4) {
#pragma unroll
        for (int b = 0; b < NB; b++) {
          const float4 hv = *reinterpret_cast<const float4*>(&hs[b][gq * KQ + i]);   // rows b >= nb: stale, never read back
          acc[b] += hv.x * wq[i] + hv.y * wq[i + 1] + hv.z * wq[i + 2] + hv.w * wq[i + 3];
        }
      }
#pragma unroll
      for (int b = 0; b < NB; b++) ored[gq][b][dh_t] = acc[b];
    }
    if (fz.Wo != nullptr) {                        // (uniform) the Wq registers are dead: request the Wo slice
      const float* wp = fz.Wo + (int64_t)(h * 64 + (actw ? jgw : 0) * RPGW) * DD + cfw * 4;
#pragma unroll
      for (int i = 0; i < RPGW; i++) wo[i] = *reinterpret_cast<const float4*>(wp + (int64_t)i * DD);
    }
    __syncthreads();
    for (int e = tid; e < nb * 64; e += 256) {
      const int b = e >> 6, dh = e & 63;
      qs[b][dh] = (((ored[0][b][dh] + ored[1][b][dh]) + (ored[2][b][dh] + ored[3][b][dh])) + bq[h * 64 + dh]) * scale;
    }
    __syncthreads();
  } else {
    // q = (x Wq + bq) * s  (mod.rs:483, :506-509)
    for (int e = tid; e < nb * 64; e += 256) {
      const int b = e >> 6, dh = e & 63, col = h * 64 + dh;
      qs[b][dh] = fold_partials(Pq, KS, (int64_t)lay.S * d, (int64_t)slots[b] * d + col, bq[col]) * scale;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int e = tid + i * 256, r = e >> 4, q4 = (e & 15) * 4;
      Kt[r][q4 + 0] = kreg[i].x; Kt[r][q4 + 1] = kreg[i].y; Kt[r][q4 + 2] = kreg[i].z; Kt[r][q4 + 3] = kreg[i].w;
    }
    __syncthreads();
  }
  {
    const int j = tid & (CA_CH - 1), hf = tid >> 7;
    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) acc[b] = 0.f;
#pragma unroll 8
    for (int dh = hf * 32; dh < hf * 32 + 32; dh++) {
      const float kv = Kt[j][dh];
#pragma unroll
      for (int b = 0; b < NB; b++) acc[b] += qs[b][dh] * kv;   // rows b >= nb hold stale q: never read back
    }
#pragma unroll
    for (int b = 0; b < NB; b++) sp[hf][b][j] = acc[b];
  }
  __syncthreads();
  {
    const int wave = tid >> 6, lane = tid & 63;
    for (int b = wave; b < nb; b += 4) {
      float s0 = lane < nk ? sp[0][b][lane] + sp[1][b][lane] : -INFINITY;
      float s1 = lane + 64 < nk ? sp[0][b][lane + 64] + sp[1][b][lane + 64] : -INFINITY;
      const float m = wave_max(fmaxf(s0, s1));
      const float e0 = lane < nk ? expf(s0 - m) : 0.f;
      const float e1 = lane + 64 < nk ? expf(s1 - m) : 0.f;
      pb[b][lane] = e0; pb[b][lane + 64] = e1;
      const float l = wave_sum(e0 + e1);
      if (lane == 0) { stat[b][0] = m; stat[b][1] = l; }
    }
  }
  __syncthreads();
  {
    float o[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) o[b] = 0.f;
#pragma unroll
    for (int i = 0; i < 32; i++) {
      const int j = gq * 32 + i;
#pragma unroll
      for (int b = 0; b < NB; b++) o[b] += (b < nb ? pb[b][j] : 0.f) * vreg[i];   // rows past nk carry p = 0 / v = 0
    }
#pragma unroll
    for (int b = 0; b < NB; b++) ored[gq][b][dh_t] = o[b];
  }
  __syncthreads();
  if (fz.Wo == nullptr) {
    for (int e = tid; e < nb * 64; e += 256) {
      const int b = e >> 6, dh = e & 63;
      float* dst = ca + ((int64_t)(slots[b] * n_head + h) * n_chunks + c) * CA_STRIDE;
      dst[2 + dh] = (ored[0][b][dh] + ored[1][b][dh]) + (ored[2][b][dh] + ored[3][b][dh]);
      if (dh == 0) { dst[0] = stat[b][0]; dst[1] = stat[b][1]; }
    }
    return;
  }
  // ---- fused out-projection: P[b][:] = o_chunk[b][:] Wo[h*64 .. h*64+64][:]  (mod.rs:489 is linear: the chunk
  // combine and the softmax normalisation commute with it and are applied by the consumer)
  for (int e = tid; e < nb * 64; e += 256) {
    const int b = e >> 6, dh = e & 63;
    qs[b][dh] = (ored[0][b][dh] + ored[1][b][dh]) + (ored[2][b][dh] + ored[3][b][dh]);   // (the queries are dead)
  }
  __syncthreads();
  if constexpr (KQ > 0) {
    float o4[NB][4];
#pragma unroll
    for (int b = 0; b < NB; b++) { o4[b][0] = o4[b][1] = o4[b][2] = o4[b][3] = 0.f; }
    const int jb = (actw ? jgw : 0) * RPGW;
#pragma unroll
    for (int i = 0; i < RPGW; i++) {
#pragma unroll
      for (int b = 0; b < NB; b++) {
        const float av = qs[b][jb + i];
        o4[b][0] += av * wo[i].x; o4[b][1] += av * wo[i].y; o4[b][2] += av * wo[i].z; o4[b][3] += av * wo[i].w;
      }
    }
    float* obuf = &Kt[0][0];                        // [GW - 1][NB][DD] floats <= 7 * 8 * 128 or 1 * 8 * 512 < 128 * 65
    if (actw && jgw > 0) {
#pragma unroll
      for (int b = 0; b < NB; b++)
        *reinterpret_cast<float4*>(&obuf[((jgw - 1) * NB + b) * DD + cfw * 4]) = make_float4(o4[b][0], o4[b][1], o4[b][2], o4[b][3]);
    }
    __syncthreads();
    if (jgw == 0) {
#pragma unroll
      for (int g2 = 1; g2 < GW; g2++) {             // group order fixed: deterministic sums
#pragma unroll
        for (int b = 0; b < NB; b++) {
          const float4 t = *reinterpret_cast<const float4*>(&obuf[((g2 - 1) * NB + b) * DD + cfw * 4]);
          o4[b][0] += t.x; o4[b][1] += t.y; o4[b][2] += t.z; o4[b][3] += t.w;
        }
      }
      // record layout: [plane = h * n_chunks + c][S][2 + d]
      constexpr int RS = DD + 2;
      float* base = fz.rec + (int64_t)(h * n_chunks + c) * lay.S * RS;
#pragma unroll
      for (int b = 0; b < NB; b++) {
        if (b < nb) {
          float* dst = base + (int64_t)slots[b] * RS;
          dst[2 + cfw * 4 + 0] = o4[b][0]; dst[2 + cfw * 4 + 1] = o4[b][1];
          dst[2 + cfw * 4 + 2] = o4[b][2]; dst[2 + cfw * 4 + 3] = o4[b][3];
          if (cfw == 0) { dst[0] = stat[b][0]; dst[1] = stat[b][1]; }
        }
      }
    }
  }
}


// ---- batch-mode cross-attention, one beam per window (greedy over many windows: configs #4 / #5) -------------
// One block per (head, window) STREAMS the window's whole cached K and V for that head (C <= 1500 keys x 64 x 2 x 4 B =
// up to 768 KB) instead of one block per 128-key chunk + a combine launch: the four waves take the four 32-key
// quarters of every 128-key chunk and keep private flash-softmax states (m, l, o) -- there is NO block barrier and no
// LDS in the loop -- which are merged once at the end.  Lane = (key row rr = lane / 16, float4 column c16 = lane % 16):
// a 16-lane DPP row reads one 256-byte head row, the 4-term partial dot products are reduced over the row with DPP,
// so the four row groups of the wave hold the scores (and probabilities) of their own keys and the P.V product needs
// no exchange either: each lane accumulates its 4 output dims over its row group's keys; the row groups are summed
// once after the loop (v_permlane16/32_swap).  Three register sets rotate so that the next chunk's K and V are in
// flight while the current one is consumed; the normalised head output goes straight to `att` (mod.rs:518-532) and
// the chunk-combine launch disappears.
//
// FUSE: the block also does what precedes the attention in the sublayer (mod.rs:345-350, :482-483) for ITS row: fold the
// self-attention out-projection's planes into the residual stream (head 0's block writes the folded row), cross_attn_ln,
// and the query projection of its head -- q = ln(x) . Wq[:, 64 h .. 64 h + 63] + bq, the 16 row groups of the block
// taking every 16th K-row of the head's weight slice (327 KB at d = 1280: L2 traffic, the slice is shared by every
// window's block of the head) and meeting in LDS in a fixed order.  The cached K/V of the first chunk are requested
// before any of it.  Two launches (dec_resolve_ln, the Wq GEMM) and their kernel boundaries disappear.
template <bool FUSE>
__global__ __launch_bounds__(256) void dec_cross_attn_stream_kernel(const int* __restrict__ st, StepLayout lay,
                                                                    const float* __restrict__ Pq, int KS,
                                                                    const float* __restrict__ bq, int d,
                                                                    const float* __restrict__ ckv, int ldkv, int koff,
                                                                    const int* __restrict__ win_row0,
                                                                    const int* __restrict__ win_C, float scale,
                                                                    float* __restrict__ att, CaStreamFuse fz) {
  __shared__ float red[4][66];                       // per wave: m, l, o[64]
  __shared__ __attribute__((aligned(16))) float hs[FUSE ? CSF_MAX_D : 1];
  __shared__ __attribute__((aligned(16))) float qred[FUSE ? 16 * 64 : 1];
  const int h = blockIdx.x, w = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = st[lay.win_nb + w];
  if (nb == 0 || st[ST_N] == 0) return;
  const int slot = st[lay.win_slots + w * MAX_BEAMS];
  const int C = win_C[w];
  const int rr = lane >> 4, c16 = lane & 15;
  const float* Kb = ckv + (int64_t)win_row0[w] * ldkv + koff + h * 64 + c16 * 4;   // K pre-scaled at projection time
  const float* Vb = Kb + d;
  const int n_chunks = (C + CA_CH - 1) / CA_CH;
  // keys past C re-read row C - 1 (always valid): every load is unconditional, the score is masked instead
  auto key_of = [&](int c, int i) { return c * CA_CH + wave * 32 + 4 * i + rr; };
  // three rotating register sets of 8 float4: while chunk c is consumed (K set -> scores, V set -> P.V), chunk c + 1's K
  // goes into the free set at the top of the step and its V into the K set as soon as the scores are done -- 16-24 KB
  // per wave stay in flight with ~130 VGPRs (three waves per SIMD: the grid has 2.4-3 blocks per CU, one round)
  float4 s0[8], s1[8], s2[8];
  auto load_k = [&](float4 (&r)[8], int c) {
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = *reinterpret_cast<const float4*>(Kb + (int64_t)min(key_of(c, i), C - 1) * ldkv);
  };
  auto load_v = [&](float4 (&r)[8], int c) {
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = *reinterpret_cast<const float4*>(Vb + (int64_t)min(key_of(c, i), C - 1) * ldkv);
  };
  load_k(s0, 0);
  load_v(s1, 0);
  // q = (x Wq + bq) * s  (mod.rs:483, :506-509): this lane's four head dims
  float q[4];
  if constexpr (FUSE) {
    // ---- x + (bias + out-projection planes), s ascending (mod.rs:346-348); LayerNorm (two passes, biased variance)
    constexpr int VPT = CSF_MAX_D / 256;
    float v[VPT];
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; i++) {
      const int c = tid + i * 256;
      v[i] = 0.f;
      if (c < d) {
        const float a = fz.x_in[(int64_t)slot * d + c] +
                        fold_partials(fz.pend, fz.KSp, (int64_t)lay.S * d, (int64_t)slot * d + c, fz.pbias[c]);
        if (h == 0) fz.x_out[(int64_t)slot * d + c] = a;
        v[i] = a;
        sm += a;
      }
    }
    sm = wave_sum(sm);
    if (lane == 0) red[0][wave] = sm;
    __syncthreads();
    const float mean = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)d;
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; i++) {
      const int c = tid + i * 256;
      if (c < d) { const float t = v[i] - mean; qq += t * t; }
    }
    qq = wave_sum(qq);
    if (lane == 0) red[1][wave] = qq;
    __syncthreads();
    const float var = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)d;
    const float denom = fz.ln_inside ? sqrtf(var + fz.ln_eps) : (sqrtf(var) + fz.ln_eps);
#pragma unroll
    for (int i = 0; i < VPT; i++) {
      const int c = tid + i * 256;
      if (c < d) hs[c] = (v[i] - mean) / denom * fz.ln_g[c] + fz.ln_b[c];
    }
    __syncthreads();
    // ---- q: row group g16 = tid / 16 takes K-rows g16, g16 + 16, ...; eight rows per round, two rounds in flight
    const int g16 = tid >> 4;
    const float* wp = fz.Wq + (int64_t)g16 * d + h * 64 + c16 * 4;
    const int nk = d >> 4;                            // K-rows per group (host: d % 128 == 0)
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float4 wa[8], wb[8];
    auto load_w = [&](float4 (&wr)[8], int i0) {
#pragma unroll
      for (int j = 0; j < 8; j++) wr[j] = *reinterpret_cast<const float4*>(wp + (int64_t)(16 * (i0 + j)) * d);
    };
    auto fma_w = [&](const float4 (&wr)[8], int i0) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const float xv = hs[g16 + 16 * (i0 + j)];
        acc[0] += xv * wr[j].x; acc[1] += xv * wr[j].y; acc[2] += xv * wr[j].z; acc[3] += xv * wr[j].w;
      }
    };
    load_w(wa, 0);
#pragma unroll 1
    for (int i0 = 0; i0 < nk; i0 += 16) {
      if (i0 + 8 < nk) load_w(wb, i0 + 8);
      fma_w(wa, i0);
      if (i0 + 16 < nk) load_w(wa, i0 + 16);
      if (i0 + 8 < nk) fma_w(wb, i0 + 8);
    }
    *reinterpret_cast<float4*>(&qred[g16 * 64 + c16 * 4]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const int col = c16 * 4 + t;
      float sq = bq[h * 64 + col];
#pragma unroll
      for (int g = 0; g < 16; g++) sq += qred[g * 64 + col];            // group order fixed
      q[t] = sq * scale;
    }
    __syncthreads();                                  // (red is reused by the merge below)
  } else {
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const int col = h * 64 + c16 * 4 + t;
    q[t] = fold_partials(Pq, KS, (int64_t)lay.S * d, (int64_t)slot * d + col, bq[col]) * scale;
  }
  }
  float m = -INFINITY, l = 0.f, o[4] = {0.f, 0.f, 0.f, 0.f};
  auto step = [&](float4 (&kr)[8], float4 (&vr)[8], float4 (&fr)[8], int c) {
    const bool more = c + 1 < n_chunks;               // (uniform)
    if (more) load_k(fr, c + 1);
    float sc[8];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const float part = (kr[i].x * q[0] + kr[i].y * q[1]) + (kr[i].z * q[2] + kr[i].w * q[3]);
      sc[i] = row16_sum(part);                        // the 16 lanes of a row now hold their key's score
      if (key_of(c, i) >= C) sc[i] = -INFINITY;
      mx = fmaxf(mx, sc[i]);
    }
    if (more) load_v(kr, c + 1);                      // the K registers are free: the next chunk's V
    mx = xor32_max(xor16_max(mx));                    // over the wave's 32 keys of this chunk
    const float m_new = fmaxf(m, mx);
    if (m_new > -INFINITY) {                          // (wave-uniform) a quarter wholly past C contributes nothing
      const float alpha = expf(m - m_new);            // m == -inf: 0
      float ps = 0.f, po[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const float p = expf(sc[i] - m_new);          // masked keys: exp(-inf) = 0
        ps += p;
        po[0] += p * vr[i].x; po[1] += p * vr[i].y; po[2] += p * vr[i].z; po[3] += p * vr[i].w;
      }
      ps = xor32_sum(xor16_sum(ps));                  // every lane of a row group holds the same p: sum the 4 groups
      l = l * alpha + ps;
#pragma unroll
      for (int t = 0; t < 4; t++) o[t] = o[t] * alpha + po[t];
      m = m_new;
    }
  };
#pragma unroll 1
  for (int c = 0; c < n_chunks; c += 3) {
    step(s0, s1, s2, c);                              // K = s0, V = s1, free = s2
    if (c + 1 < n_chunks) step(s2, s0, s1, c + 1);    // K = s2, V = s0, free = s1
    if (c + 2 < n_chunks) step(s1, s2, s0, c + 2);    // K = s1, V = s2, free = s0
  }
#pragma unroll
  for (int t = 0; t < 4; t++) o[t] = xor32_sum(xor16_sum(o[t]));   // the four row groups' partial outputs
  if (rr == 0) {
#pragma unroll
    for (int t = 0; t < 4; t++) red[wave][2 + c16 * 4 + t] = o[t];
  }
  if (lane == 0) { red[wave][0] = m; red[wave][1] = l; }
  __syncthreads();
  if (tid < 64) {                                     // merge the four waves' states in a fixed order
    const float M = fmaxf(fmaxf(red[0][0], red[1][0]), fmaxf(red[2][0], red[3][0]));
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float wgt = expf(red[j][0] - M);
      num += wgt * red[j][2 + tid];
      den += wgt * red[j][1];
    }
    att[(int64_t)slot * d + h * 64 + tid] = num / den;
  }
}

// ---- batch mode (more than 8 live beams): prologues materialised for the split-K MFMA GEMM ------
// out[r][k] = GELU(bias[k] + sum_s P[s][r][k])   (mod.rs:377-378)
__global__ void dec_gelu_fold_kernel(const int* __restrict__ st, const float* __restrict__ P, int KS, int S, int K,
                                     const float* __restrict__ bias, float* __restrict__ out) {
  const int r = blockIdx.y;
  if (r >= st[ST_N]) return;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  out[(int64_t)r * K + k] = gelu_erf(fold_partials(P, KS, (int64_t)S * K, (int64_t)r * K + k, bias[k]));
}
// out[r][h*64+dh] = combine of the cross-attention key-chunk partials
__global__ void dec_attn_combine_kernel(const int* __restrict__ st, const float* __restrict__ ca, int n_head,
                                        int n_chunks, float* __restrict__ out) {
  const int r = blockIdx.y;
  if (r >= st[ST_N]) return;
  const int k = blockIdx.x * blockDim.x + threadIdx.x, d = n_head * 64;
  if (k >= d) return;
  const int hh = k >> 6, dh = k & 63;
  out[(int64_t)r * d + k] = combine_chunks(ca + ((int64_t)(r * n_head + hh) * n_chunks) * CA_STRIDE, n_chunks, dh);
}

// mask + log_softmax + top-k of one beam's full logits row (batch mode; transcribe.rs:271-304)
__global__ __launch_bounds__(1024) void dec_topk_rows_kernel(int* __restrict__ st, const float* __restrict__ logits,
                                                              int V, const float* __restrict__ mask, int use_mask,
                                                              int k, int32_t* __restrict__ out_id,
                                                              float* __restrict__ out_lp, float* __restrict__ row_stats,
                                                              StepLayout lay, int* __restrict__ gctl,
                                                              int* __restrict__ gtok, int Lmax, int eot) {
  __shared__ float redv[16];
  __shared__ int redi[16];
  __shared__ float bc[2];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (r >= st[ST_N]) return;
  const float* x = logits + (int64_t)r * V;
  float tv[TOPK_MAX];
  int ti[TOPK_MAX];
#pragma unroll
  for (int j = 0; j < TOPK_MAX; j++) { tv[j] = -INFINITY; ti[j] = 0x7fffffff; }
  // Both passes walk the row in batches of TR_PF values per thread: the loads of a batch are requested together, then
  // consumed in ascending column order -- the arithmetic and its order are those of the one-value-at-a-time loop (bit-
  // identical results), but a pass is ~9 memory round trips instead of 51 (the kernel was 70 - 75 us per step for 15 - 38
  // rows, all of it load latency: profiles/r05_b_bench_default.json).
  constexpr int TR_PF = 6;
  float m = -INFINITY;
  for (int c0 = tid; c0 < V; c0 += 1024 * TR_PF) {
    float vb[TR_PF], mb[TR_PF];
#pragma unroll
    for (int u = 0; u < TR_PF; u++) {
      const int c = c0 + 1024 * u, cc = c < V ? c : tid;
      vb[u] = x[cc];
      mb[u] = use_mask ? mask[cc] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < TR_PF; u++) {
      const int c = c0 + 1024 * u;
      if (c < V) {
        float v = vb[u];
        if (use_mask) v += mb[u];
        m = fmaxf(m, v);
        if (v > tv[TOPK_MAX - 1]) {   // ids arrive ascending per thread: equal values never displace
          float cv = v; int ci = c;
#pragma unroll
          for (int j = 0; j < TOPK_MAX; j++)
            if (cv > tv[j]) { float t = tv[j]; int w = ti[j]; tv[j] = cv; ti[j] = ci; cv = t; ci = w; }
        }
      }
    }
  }
  m = wave_max(m);
  if (lane == 0) redv[wave] = m;
  __syncthreads();
  float M = redv[0];
  for (int j = 1; j < 16; j++) M = fmaxf(M, redv[j]);
  float s = 0.f;
  for (int c0 = tid; c0 < V; c0 += 1024 * TR_PF) {
    float vb[TR_PF], mb[TR_PF];
#pragma unroll
    for (int u = 0; u < TR_PF; u++) {
      const int c = c0 + 1024 * u, cc = c < V ? c : tid;
      vb[u] = x[cc];
      mb[u] = use_mask ? mask[cc] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < TR_PF; u++) {
      if (c0 + 1024 * u < V) {
        float v = vb[u];
        if (use_mask) v += mb[u];
        s += expf(v - M);
      }
    }
  }
  s = wave_sum(s);
  __syncthreads();
  if (lane == 0) redv[wave] = s;
  __syncthreads();
  if (tid == 0) {
    float ss = 0.f;
    for (int j = 0; j < 16; j++) ss += redv[j];
    bc[0] = logf(ss);
    row_stats[2 * r] = M; row_stats[2 * r + 1] = bc[0];
  }
  __syncthreads();
  const float lse = bc[0];
  for (int round = 0; round < k; round++) {
    float bv = tv[0]; int bi = ti[0];
    wave_argmax(bv, bi);
    __syncthreads();
    if (lane == 0) { redv[wave] = bv; redi[wave] = bi; }
    __syncthreads();
    float gv = redv[0]; int gi = redi[0];
    for (int j = 1; j < 16; j++)
      if (better(redv[j], redi[j], gv, gi)) { gv = redv[j]; gi = redi[j]; }
    if (tid == 0) {
      const int tok = (gctl && round == 0) ? top1_or_eot(gi, eot, gctl) : gi;
      out_id[r * TOPK_MAX + round] = tok;
      out_lp[r * TOPK_MAX + round] = (gv - M) - lse;
      if (gctl && round == 0)
        chained_update(st, lay, gctl, gtok, Lmax, eot, r, tok, st[lay.len + r], gctl[GC_HDR + lay.S + r], st[ST_STEP]);
    }
    if (ti[0] == gi) {
#pragma unroll
      for (int j = 0; j < TOPK_MAX - 1; j++) { tv[j] = tv[j + 1]; ti[j] = ti[j + 1]; }
      tv[TOPK_MAX - 1] = -INFINITY; ti[TOPK_MAX - 1] = 0x7fffffff;
    }
  }
}

// One WAVE per window restates transcribe.cpp's beam_search_windows loop body (which restates beam.rs) on the device.
//
// beam.rs's get_top_elements is a streaming insertion (ascending list; insert before the first stored score >= the new one;
// evict index 0 when over capacity; skip when full and score < min).  Its result is a function of the sequence only through
// a total order: the survivors are the `num` best by (score descending, position in the sequence ascending) -- on a tie at
// the boundary the later element is inserted at index 0 and evicted at once, so the earlier one survives -- and the list ends
// up ascending by score with the LATER of two equal scores in front.  Hence, with rank = #elements that beat an element in
// that order, element e survives iff rank < num and sits at index (n_kept - 1 - rank).  That is what the lanes compute, one
// candidate each, instead of replaying the insertion serially (the first version of this kernel did, one thread per window:
// 74 us per step at tiny.en, all of it dependent memory round trips and scratch traffic -- profiles/r06_b_bench.json).
// A beam's k continuations enter the sequence sorted by token id (transcribe.cpp / transcribe.rs:286-304 produce them in id
// order), and all k survive the per-beam pass (it holds exactly k), ordered by that same rule.  Scores are f64:
// log_prob + (double)lp (transcribe.rs:299).  max_by takes the LAST of equal maxima (beam.rs:23-27, :33-36).
__global__ __launch_bounds__(1024) void dec_beam_update_kernel(BeamChainArgs a) {
  constexpr int NW = 16;                              // waves per block = windows per pass
  __shared__ int w_node[NW][BEAM_KB], w_fin[NW][BEAM_KB], w_prev[NW][BEAM_KB], w_slot[NW][BEAM_KB];
  __shared__ double w_lp[NW][BEAM_KB];
  __shared__ int c_tok[NW][64], c_seq[NW][64];
  __shared__ double c_sc[NW][64];
  __shared__ int o_node[NW][BEAM_KB], o_fin[NW][BEAM_KB], o_prev[NW][BEAM_KB], o_tok[NW][BEAM_KB];
  __shared__ double o_lp[NW][BEAM_KB];
  // what the state block of the next step needs, kept for every window of the batch (<= 64)
  __shared__ int g_tok[64][BEAM_KB], g_fin[64][BEAM_KB], g_prev[64][BEAM_KB];
  __shared__ int g_nb[64], g_done[64], n_live_w[64], base_w[64];
  __shared__ int s_tok[64 * MAX_BEAMS], s_par[64 * MAX_BEAMS];      // the next step's rows by slot (for the prepare phase)
  __shared__ int n_total;
  const BeamChainLayout& B = a.bl;
  const StepLayout& L = a.lay;
  int* ctl = a.ctl;
  double* lpv = reinterpret_cast<double*>(ctl + B.lp);
  int2* nodes = reinterpret_cast<int2*>(ctl + B.nodes);
  const int W = B.W, k = a.k;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int depth = ctl[BC_DEPTH];                    // decode steps completed before this call's step
  for (int w0 = 0; w0 < W; w0 += NW) {                // (block-uniform trip count: every wave meets every barrier)
    const int w = w0 + wv;
    const bool act = w < W;
    const int nb = act ? ctl[B.nb + w] : 0;
    const int was_done = act ? ctl[B.done + w] : 1;
    // ---- the window's beams: one round trip ----
    int my_fin = 0; double my_lp = 0.0;
    if (lane < BEAM_KB) {
      int nd = -1, fn = 0, pv = -1, sl = -1; double lp = 0.0;
      if (act) {
        // (requested together with nb / done above -- entries past nb hold zeros or an earlier generation: masked below)
        const int o = w * BEAM_KB + lane;
        nd = ctl[B.node + o]; fn = ctl[B.fin + o]; pv = ctl[B.prev_slot + o]; sl = ctl[B.slot_now + o]; lp = lpv[o];
        if (lane >= nb) { nd = -1; fn = 0; pv = -1; sl = -1; lp = 0.0; }
      }
      w_node[wv][lane] = nd; w_fin[wv][lane] = fn; w_prev[wv][lane] = pv; w_slot[wv][lane] = sl; w_lp[wv][lane] = lp;
      my_fin = fn; my_lp = lp;
    }
    const bool stepA = act && !a.first && !was_done;
    // order of the unfinished / finished beams among themselves (list order)
    const unsigned long long unf_mask = __ballot(lane < nb && !my_fin), fin_mask = __ballot(lane < nb && my_fin);
    const int n_unf = __popcll(unf_mask), n_fin = __popcll(fin_mask);
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    __syncthreads();
    // ---- candidates: lane c = (b_ord, j) -> the j-th top-k entry of the b_ord-th unfinished beam ----
    int tok = -1, src = -1; double sc = 0.0; bool cand = false;
    if (stepA && lane < n_unf * k) {
      const int b_ord = lane / k, j = lane - b_ord * k;
      // the b_ord-th set bit of unf_mask
      unsigned long long mm = unf_mask; for (int t = 0; t < b_ord; t++) mm &= mm - 1;
      src = __ffsll((long long)mm) - 1;
      const int slot = w_slot[wv][src];
      tok = a.topk_id[slot * TOPK_MAX + j];
      sc = w_lp[wv][src] + (double)a.topk_lp[slot * TOPK_MAX + j];      // transcribe.rs:299
      // a row without k finite candidates (NaN logits) leaves 0x7fffffff in its list: never an embedding index -- the host
      // fails the call (BC_ERR); distinct negative stand-ins keep the id order below well-defined
      if ((unsigned)tok >= (unsigned)a.V) { ctl[BC_ERR] = 1; tok = -1 - j; }
      if (!(sc == sc)) ctl[BC_ERR] = 1;
      cand = true;
    }
    c_tok[wv][lane] = tok; c_sc[wv][lane] = sc;
    __syncthreads();
    int seq = 0x7fffffff;
    if (cand) {
      // position inside the beam's own pass: ascending score, the later (larger id) of two equal scores in front
      const int b_ord = lane / k;
      int pos = 0;
      for (int j2 = 0; j2 < k; j2++) {
        const double s2 = c_sc[wv][b_ord * k + j2]; const int t2 = c_tok[wv][b_ord * k + j2];
        if (s2 < sc || (s2 == sc && t2 > tok)) pos++;
      }
      seq = b_ord * k + pos;
    }
    c_seq[wv][lane] = seq;
    __syncthreads();
    const int n_cand = stepA ? n_unf * k : 0;
    const int ns = min(k, n_cand), nf = stepA ? min(k, n_fin) : 0;
    if (cand) {
      int rank = 0;
      for (int c2 = 0; c2 < n_cand; c2++) {
        const double s2 = c_sc[wv][c2];
        if (s2 > sc || (s2 == sc && c_seq[wv][c2] < seq)) rank++;
      }
      if (rank < k) {
        const int q = ns - 1 - rank;
        const int tk = tok < 0 ? a.eot : tok;
        const int pool = (depth + 1) * W * BEAM_KB + w * BEAM_KB + q;
        nodes[pool] = make_int2(tk, w_node[wv][src]);
        o_node[wv][q] = pool; o_fin[wv][q] = tk == a.eot ? 1 : 0; o_prev[wv][q] = w_slot[wv][src]; o_lp[wv][q] = sc; o_tok[wv][q] = tk;
      }
    }
    if (stepA && lane < nb && my_fin) {                 // finished beams: carried, the best k of them (beam.rs:56-57, :75-78)
      const int f_ord = __popcll(fin_mask & below);
      int rank = 0;
      for (int i2 = 0; i2 < nb; i2++) {
        if (!w_fin[wv][i2] || i2 == lane) continue;
        const double s2 = w_lp[wv][i2];
        const int f2 = __popcll(fin_mask & (i2 == 0 ? 0ull : (~0ull >> (64 - i2))));
        if (s2 > my_lp || (s2 == my_lp && f2 < f_ord)) rank++;
      }
      if (rank < k) {
        const int q = ns + (nf - 1 - rank);
        o_node[wv][q] = w_node[wv][lane]; o_fin[wv][q] = 1; o_prev[wv][q] = w_prev[wv][lane]; o_lp[wv][q] = my_lp; o_tok[wv][q] = a.eot;
      }
    }
    __syncthreads();
    // ---- the next generation (or, with no step behind this call / an ended window, the current one) ----
    const int n_next = stepA ? ns + nf : nb;
    int cur_fin = 0; double cur_lp = 0.0;
    if (act && lane < BEAM_KB) {
      int nd, fn, pv, tk; double lp;
      if (stepA) { nd = o_node[wv][lane]; fn = o_fin[wv][lane]; pv = o_prev[wv][lane]; lp = o_lp[wv][lane]; tk = o_tok[wv][lane]; }
      else { nd = w_node[wv][lane]; fn = w_fin[wv][lane]; pv = w_prev[wv][lane]; lp = w_lp[wv][lane]; tk = (lane < nb && nd >= 0) ? nodes[nd].x : 0; }
      if (lane < n_next) {
        if (stepA) {
          const int o = w * BEAM_KB + lane;
          ctl[B.node + o] = nd; ctl[B.fin + o] = fn; ctl[B.prev_slot + o] = pv; lpv[o] = lp;
        }
        cur_fin = fn; cur_lp = lp;
        if (!(lp == lp)) ctl[BC_ERR] = 1;             // NaN log-probability: the reference panics (partial_cmp().unwrap())
      }
      g_tok[w][lane] = tk; g_fin[w][lane] = fn; g_prev[w][lane] = pv;
    }
    // ---- termination test of the next iteration (beam.rs:23-27): the LAST of the equal maxima ----
    double best_lp = cur_lp; int best_i = (act && lane < n_next) ? lane : -1;
#pragma unroll
    for (int off = 1; off < BEAM_KB; off <<= 1) {      // butterfly over the first BEAM_KB lanes (the rest hold -1)
      const double o_l = __shfl_xor(best_lp, off); const int o_i = __shfl_xor(best_i, off);
      if (o_i >= 0 && (best_i < 0 || o_l > best_lp || (o_l == best_lp && o_i > best_i))) { best_lp = o_l; best_i = o_i; }
    }
    best_i = __shfl(best_i, 0);
    const int best_fin = best_i >= 0 ? __shfl(cur_fin, best_i) : 0;
    const int now_done = was_done || (best_i >= 0 && best_fin);
    const int live = now_done ? 0 : __popcll(__ballot(lane < n_next && !cur_fin));
    if (act && lane == 0) {
      if (stepA) ctl[B.nb + w] = n_next;
      if (now_done && !was_done) ctl[B.done + w] = 1;
      g_nb[w] = n_next; g_done[w] = now_done; n_live_w[w] = live;
    }
    __syncthreads();
  }
  if (tid == 0) {
    int acc = 0;
    for (int i = 0; i < W; i++) { base_w[i] = acc; acc += n_live_w[i]; }
    const int last = !a.first && depth + 1 >= B.max_depth;      // the step that just ran was the last one allowed
    if (last) acc = 0;
    if (!a.first) ctl[BC_DEPTH] = depth + 1;
    ctl[BC_NLIVE] = acc;
    if (acc == 0) { ctl[BC_ALLDONE] = 1; if (a.done_flag_host) __hip_atomic_store(a.done_flag_host, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    n_total = acc;
  }
  __syncthreads();
  const int n_all = n_total;
  // ---- the next step's state block (what wb_session_step writes on the host) ----
  const int step_next = a.step_pos + depth + (a.first ? 0 : 1);
  int* so = a.state_out;
  for (int e = tid; e < L.total; e += 1024) {
    int v = 0;
    if (e == ST_N) v = n_all;
    else if (e == ST_STEP) v = step_next;
    so[e] = v;
  }
  __syncthreads();
  if (n_all > 0) {
    for (int w = wv; w < W; w += NW) {
      const int nb = g_nb[w];
      const bool is_done = g_done[w] != 0;
      const bool lv = lane < nb && !is_done && !g_fin[w][lane];
      const unsigned long long lm = __ballot(lv);
      const int j = __popcll(lm & (lane == 0 ? 0ull : (~0ull >> (64 - lane))));
      int slot = -1;
      if (lv) {
        slot = base_w[w] + j;
        so[L.tok + slot] = g_tok[w][lane]; so[L.parent + slot] = g_prev[w][lane]; so[L.len + slot] = step_next + 1; so[L.win + slot] = w;
        so[L.win_slots + w * MAX_BEAMS + j] = slot;
        s_tok[slot] = g_tok[w][lane]; s_par[slot] = g_prev[w][lane];
      }
      if (lane < nb) ctl[B.slot_now + w * BEAM_KB + lane] = slot;
      if (lane == 0) so[L.win_nb + w] = is_done ? 0 : __popcll(lm);
    }
  }
  // ---- the next step's rows: position tables (copied from the parent's, double-buffered by step parity) and
  // x = E[token] + pos[len - 1] (mod.rs:141-146) -- what dec_prepare_kernel does for a host-driven step
  if (a.x != nullptr && n_all > 0) {
    __syncthreads();                                  // the state block above is complete
    int* tab_new = a.tabs + (size_t)(step_next & 1) * L.S * a.Lmax;
    const int* tab_old = a.tabs + (size_t)((step_next & 1) ^ 1) * L.S * a.Lmax;
    const int len = step_next + 1, d4 = a.d >> 2;
    for (int i = wv; i < n_all; i += NW) {            // one wave per row
      const int parent = s_par[i], tk = s_tok[i];
      if (parent >= 0)
        for (int p = lane; p < len - 1; p += 64) tab_new[i * a.Lmax + p] = tab_old[parent * a.Lmax + p];
      if (lane == 0) tab_new[i * a.Lmax + len - 1] = step_next * L.S + i;
      const float4* e = reinterpret_cast<const float4*>(a.E + (int64_t)tk * a.d);
      const float4* pp = reinterpret_cast<const float4*>(a.pos + (int64_t)(len - 1) * a.d);
      float4* o = reinterpret_cast<float4*>(a.x + (int64_t)i * a.d);
      for (int c = lane; c < d4; c += 64) {
        const float4 u = e[c], v = pp[c];
        o[c] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
      }
    }
  }
}

__global__ void dec_logprob_row_kernel(const float* __restrict__ x, int KS, int64_t plane, int V,
                                       const float* __restrict__ mask, int use_mask,
                                       const float* __restrict__ stats, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= V) return;
  float v = x[c];
  for (int s2 = 1; s2 < KS; s2++) v += x[(int64_t)s2 * plane + c];
  if (use_mask) v += mask[c];
  out[c] = (v - stats[0]) - stats[1];
}

}  // namespace

void launch_dec_prepare(hipStream_t st, const int* state_host_mapped, int* state_dev, const StepLayout& lay, int n,
                        int* tabs, int Lmax, const float* E, const float* pos, int d, float* x, const int* gctl) {
  WB_KLAUNCH(dec_prepare_kernel, dim3(n), dim3(128), 0, st, state_host_mapped, state_dev, lay, tabs, Lmax, E, pos,
                     d, x, gctl);
}

void launch_dec_resolve_ln(hipStream_t st, const int* state, int n_max, const float* x_in, float* x_out,
                           const float* P, int KS, int S, const float* bias, int d, const LayerNormW& ln,
                           int eps_inside_sqrt, float* h) {
#define WB_RESOLVE(VPT_)                                                                                                \
  WB_KLAUNCH(dec_resolve_ln_kernel<VPT_>, dim3(n_max), dim3(256), 0, st, state, x_in, x_out, P, KS, S, bias, d, ln.g, ln.b, \
             ln.eps, eps_inside_sqrt, h)
  if (d <= 768) WB_RESOLVE(3);            // (columns per thread: every one of them is a real load per plane)
  else if (d <= 1024) WB_RESOLVE(4);
  else if (d <= 1280) WB_RESOLVE(5);
  else WB_RESOLVE(8);
#undef WB_RESOLVE
}

void gemv_plan(int K, int N, int* KS, int* KSL) {
  // slices of >= 64 rows, <= GV_KSL_MAX, at most KS_MAX partials, aiming at >= ~512 blocks
  const int tiles = (N + GV_CT - 1) / GV_CT;
  int ks = std::max(1, std::min(KS_MAX, 512 / std::max(tiles, 1)));
  ks = std::min(ks, std::max(1, K / 64));
  int ksl = ((K + ks - 1) / ks + 31) / 32 * 32;
  if (ksl > GV_KSL_MAX) ksl = GV_KSL_MAX;
  ks = (K + ksl - 1) / ksl;
  *KS = ks; *KSL = ksl;
}

template <typename K>
static void launch_gemv_k(K kernel, hipStream_t st, dim3 grid, const GemvArgs& a) {
  WB_KLAUNCH(kernel, grid, dim3(256), 0, st, a);
}

template <int MR, int XLD, bool LN, bool STATS>
static void launch_gemv_dpl(hipStream_t st, dim3 grid, const GemvArgs& a) {
  constexpr int CT = STATS ? GV_CT_LOGITS : GV_CT;
  if (!LN) { launch_gemv_k(dec_gemv_kernel<MR, XLD, 1, LN, STATS, CT>, st, grid, a); return; }
  if (a.K <= 384) launch_gemv_k(dec_gemv_kernel<MR, XLD, LN ? 6 : 1, LN, STATS, CT>, st, grid, a);
  else if (a.K <= 512) launch_gemv_k(dec_gemv_kernel<MR, XLD, LN ? 8 : 1, LN, STATS, CT>, st, grid, a);
  else if (a.K <= 768) launch_gemv_k(dec_gemv_kernel<MR, XLD, LN ? 12 : 1, LN, STATS, CT>, st, grid, a);
  else if (a.K <= 1024) launch_gemv_k(dec_gemv_kernel<MR, XLD, LN ? 16 : 1, LN, STATS, CT>, st, grid, a);
  else launch_gemv_k(dec_gemv_kernel<MR, XLD, LN ? 20 : 1, LN, STATS, CT>, st, grid, a);
}

static bool logits_mfma_enabled() {
  static const bool on = []() { const char* e = getenv("WHISPER_HIP_LOGITS_MFMA"); return !(e && e[0] == '0'); }();
  return on;
}
static bool logits_preln_enabled() {
  static const bool on = []() { const char* e = getenv("WHISPER_HIP_LOGITS_PRELN"); return !(e && e[0] == '0'); }();
  return on;
}
static bool logits_mr16_enabled() {
  static const bool on = []() { const char* e = getenv("WHISPER_HIP_LOGITS_MR16"); return !(e && e[0] == '0'); }();
  return on;
}
// the 9 - 16-row logits pass as TWO launches (fold + LayerNorm once, then the MFMA product)?  The caller tags its first launch
// accordingly (profiling classes KC_FOLD_LN_ROWS / KC_LOGITS).
bool dec_logits_two_launches(const GemvArgs& a, int n_rows_hint) {
  const int ct = a.ct > 0 ? a.ct : GV_CT_LOGITS;
  return n_rows_hint > 8 && n_rows_hint <= 16 && a.K <= 512 && logits_mr16_enabled() && logits_mfma_enabled() && a.KS == 1 &&
         a.K % 64 == 0 && a.K >= 256 && ct == GV_CT_LOGITS && logits_preln_enabled() && a.h_tmp != nullptr;
}

void launch_dec_gemv(hipStream_t st, const GemvArgs& a, int n_rows_hint, bool stats) {
  const int ct = a.ct > 0 ? a.ct : (stats ? GV_CT_LOGITS : GV_CT);
  dim3 grid((a.N + ct - 1) / ct, a.KS, n_rows_hint > 8 ? (n_rows_hint + 7) / 8 : 1);   // (z: row groups of 8)
  const bool ln = a.pro == PRO_LN;
  if (stats && n_rows_hint > 8 && n_rows_hint <= 16 && a.K <= 512 && logits_mr16_enabled()) {
    // 9 - 16 live rows on the fused sublayer path (d <= 512): ONE 16-row pass -- two row groups of 8 stream E^T twice
    // (52.7 us per step for tiny.en's 80 MB against 18 us at <= 8 rows: profiles/r05_d_beam5_fused16.txt)
    grid.z = 1;
    // ... on the matrix cores (exact-f32 MFMA, dec_logits_mfma16_kernel) unless WHISPER_HIP_LOGITS_MFMA=0
    if (logits_mfma_enabled() && a.KS == 1 && a.K % 64 == 0 && a.K >= 256 && ct == GV_CT_LOGITS && a.pro == PRO_LN) {
      if (logits_preln_enabled() && a.h_tmp) {
        // (profiling: the caller tagged this launch as KC_FOLD_LN_ROWS -- dec_logits_two_launches -- and the product
        // launch behind it carries the logits class)
        if (a.K <= 384) WB_KLAUNCH(dec_fold_ln_rows_kernel<6>, dim3(16), dim3(64), 0, st, a);
        else WB_KLAUNCH(dec_fold_ln_rows_kernel<8>, dim3(16), dim3(64), 0, st, a);
        prof_tag(KC_LOGITS, 4.0 * (double)a.N * a.K + 4.0 * 16 * ((double)a.K + a.N));
        if (a.K <= 384) launch_gemv_k(dec_logits_mfma16_kernel<6, true>, st, grid, a);
        else launch_gemv_k(dec_logits_mfma16_kernel<8, true>, st, grid, a);
        return;
      }
      if (a.K <= 384) launch_gemv_k(dec_logits_mfma16_kernel<6, false>, st, grid, a);
      else launch_gemv_k(dec_logits_mfma16_kernel<8, false>, st, grid, a);
      return;
    }
    if (a.K <= 384) launch_gemv_k(dec_gemv_kernel<16, 512, 6, true, true, GV_CT_LOGITS>, st, grid, a);
    else launch_gemv_k(dec_gemv_kernel<16, 512, 8, true, true, GV_CT_LOGITS>, st, grid, a);
    return;
  }
  if (stats) {   // logits: LN prologue over whole rows + tile statistics; rows chunked by <= 8
    if (n_rows_hint <= 4) launch_gemv_dpl<4, DMAX, true, true>(st, grid, a);
    else launch_gemv_dpl<8, DMAX, true, true>(st, grid, a);
  } else if (ln) {
    if (n_rows_hint <= 4) launch_gemv_dpl<4, GV_KSL_MAX, true, false>(st, grid, a);
    else launch_gemv_dpl<8, GV_KSL_MAX, true, false>(st, grid, a);
  } else {
    if (n_rows_hint <= 4) launch_gemv_dpl<4, GV_KSL_MAX, false, false>(st, grid, a);
    else launch_gemv_dpl<8, GV_KSL_MAX, false, false>(st, grid, a);
  }
}

void launch_dec_self_attn(hipStream_t st, const int* state, const StepLayout& lay, int n_max, int n_head,
                          const float* Pqkv, int KS, const float* bqkv, int d, float* Kc, float* Vc, const int* tab,
                          int Lmax, float scale, float* att) {
  WB_KLAUNCH(dec_self_attn_kernel, dim3(n_max, n_head), dim3(256), 0, st, state, lay, Pqkv, KS, bqkv, d, Kc, Vc,
                     tab, Lmax, scale, att);
}

bool cross_attn_can_fuse_q(int d) { return d == 128 || d == 384 || d == 512; }   // (128: the test fixtures)

void launch_dec_cross_attn(hipStream_t st, const int* state, const StepLayout& lay, int n_windows, int n_head,
                           int n_chunks, const float* Pq, int KS, const float* bq, int d, const float* ckv, int ldkv,
                           int koff, const int* win_row0, const int* win_C, float scale, float* ca, int max_nb,
                           const CaFuse* fuse) {
  dim3 grid(n_chunks, n_head, n_windows);
  const CaFuse fz = fuse ? *fuse : CaFuse();
#define WB_CA(NB_, KQ_)                                                                                              \
  WB_KLAUNCH((dec_cross_attn_kernel<NB_, KQ_>), grid, dim3(256), 0, st, state, lay, Pq, KS, bq, d, ckv, ldkv, \
                     koff, win_row0, win_C, scale, n_head, n_chunks, ca, fz)
#define WB_CA_NB(KQ_)                  \
  if (max_nb <= 1) WB_CA(1, KQ_);      \
  else if (max_nb <= 2) WB_CA(2, KQ_); \
  else if (max_nb <= 4) WB_CA(4, KQ_); \
  else WB_CA(8, KQ_)
  if (fuse && d == 128) { WB_CA_NB(32); }
  else if (fuse && d == 384) { WB_CA_NB(96); }
  else if (fuse && d == 512) { WB_CA_NB(128); }
  else { WB_CA_NB(0); }
#undef WB_CA_NB
#undef WB_CA
}

void launch_dec_cross_attn_stream(hipStream_t st, const int* state, const StepLayout& lay, int n_windows, int n_head,
                                  const float* Pq, int KS, const float* bq, int d, const float* ckv, int ldkv, int koff,
                                  const int* win_row0, const int* win_C, float scale, float* att) {
  WB_KLAUNCH(dec_cross_attn_stream_kernel<false>, dim3(n_head, n_windows), dim3(256), 0, st, state, lay, Pq, KS, bq, d, ckv,
             ldkv, koff, win_row0, win_C, scale, att, CaStreamFuse());
}

bool cross_stream_can_fuse(int d) { return d % 128 == 0 && d <= CSF_MAX_D; }

void launch_dec_cross_attn_stream_fused(hipStream_t st, const int* state, const StepLayout& lay, int n_windows, int n_head,
                                        const float* bq, int d, const float* ckv, int ldkv, int koff, const int* win_row0,
                                        const int* win_C, float scale, float* att, const CaStreamFuse& fz) {
  WB_KLAUNCH(dec_cross_attn_stream_kernel<true>, dim3(n_head, n_windows), dim3(256), 0, st, state, lay, nullptr, 0, bq, d,
             ckv, ldkv, koff, win_row0, win_C, scale, att, fz);
}

void launch_dec_topk_merge(hipStream_t st, int* state, int n_max, const float* tstats, int n_tiles, int k,
                           int32_t* out_id, float* out_lp, float* row_stats, const StepLayout& lay, int* gctl, int* gtok,
                           int Lmax, int eot, const NextPrep& nx) {
  WB_KLAUNCH(dec_topk_merge_kernel, dim3(n_max), dim3(256), 0, st, state, tstats, n_tiles, k, out_id, out_lp,
                     row_stats, lay, gctl, gtok, Lmax, eot, nx);
}

void launch_dec_beam_update(hipStream_t st, const BeamChainArgs& a) {
  WB_KLAUNCH(dec_beam_update_kernel, dim3(1), dim3(1024), 0, st, a);
}

void launch_dec_logprob_row(hipStream_t st, const float* x, int KS, int64_t plane, int V, const float* mask,
                            int use_mask, const float* stats, float* out) {
  WB_KLAUNCH(dec_logprob_row_kernel, dim3((V + 255) / 256), dim3(256), 0, st, x, KS, plane, V, mask, use_mask,
                     stats, out);
}

void launch_dec_gelu_fold(hipStream_t st, const int* state, int n_max, const float* P, int KS, int S, int K,
                          const float* bias, float* out) {
  WB_KLAUNCH(dec_gelu_fold_kernel, dim3((K + 255) / 256, n_max), dim3(256), 0, st, state, P, KS, S, K, bias, out);
}
void launch_dec_attn_combine(hipStream_t st, const int* state, int n_max, const float* ca, int n_head, int n_chunks,
                             float* out) {
  WB_KLAUNCH(dec_attn_combine_kernel, dim3((n_head * 64 + 255) / 256, n_max), dim3(256), 0, st, state, ca, n_head,
                     n_chunks, out);
}
void launch_dec_topk_rows(hipStream_t st, int* state, int n_max, const float* logits, int V, const float* mask,
                          int use_mask, int k, int32_t* out_id, float* out_lp, float* row_stats, const StepLayout& lay,
                          int* gctl, int* gtok, int Lmax, int eot) {
  WB_KLAUNCH(dec_topk_rows_kernel, dim3(n_max), dim3(1024), 0, st, state, logits, V, mask, use_mask, k, out_id,
                     out_lp, row_stats, lay, gctl, gtok, Lmax, eot);
}

int cross_attn_chunk() { return CA_CH; }

}  // namespace wb
