// K7/K8: KV-cached single-token decode step (HBM-bound weight / KV streaming).
//
// The reference re-runs the whole decoder over the whole prefix for every beam at every step
// and re-projects the cross-attention K/V in every layer (/root/reference/src/transcribe.rs:270,
// src/model/mod.rs:482-490), then ships [n, L, V] logits to log_softmax and V floats per beam to
// the host (transcribe.rs:276-284).  Here a step touches each decoder weight once:
//   skinny GEMMs (rows = live beams) stream W[K][N] with K split over blocks into deterministic
//   partial sums; consumers fold the partials (+bias, residual, LayerNorm / GELU / q-scale) in
//   their prologues, so no reduction kernel or atomics are needed;
//   self-attention reads a paged self-KV cache through per-beam position tables (beam
//   re-indexing = copying a row of ints);  cross-attention streams each window's cached K/V once
//   for all of that window's beams, split over key chunks (flash-decoding), combined by the
//   out-projection's prologue;
//   the tied-embedding logits stream E^T [d][V], then one block per beam does mask +
//   log-softmax + top-k (value descending, id ascending) so only k pairs go back to the host.
#include <hip/hip_runtime.h>

#include "decode.h"

namespace wb {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// ---- step prepare: position tables + token/position embedding (mod.rs:141-146) ----------
__global__ void dec_prepare_kernel(const int* __restrict__ st, StepLayout lay, const int* __restrict__ tab_old,
                                   int* __restrict__ tab_new, int Lmax, const float* __restrict__ E,
                                   const float* __restrict__ pos, int d, float* __restrict__ x) {
  const int i = blockIdx.x;
  if (i >= st[ST_N]) return;
  const int len = st[lay.len + i], parent = st[lay.parent + i], tok = st[lay.tok + i];
  const int step = st[ST_STEP];
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

// ---- residual resolve + LayerNorm: x += bias + sum_s P[s]; h = LN(x) -----------------------
template <int VPT>
__global__ __launch_bounds__(256) void dec_resolve_ln_kernel(const int* __restrict__ st, float* __restrict__ x,
                                                             const float* __restrict__ P, int KS, int S,
                                                             const float* __restrict__ bias, int d,
                                                             const float* __restrict__ g,
                                                             const float* __restrict__ b, float eps,
                                                             int eps_inside_sqrt, float* __restrict__ h) {
  __shared__ float red[8];
  const int r = blockIdx.x;
  if (r >= st[ST_N]) return;
  const int tid = threadIdx.x;
  float v[VPT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; i++) {
    const int c = tid + i * 256;
    v[i] = 0.f;
    if (c < d) {
      float a = x[(int64_t)r * d + c];
      if (KS > 0) {
        float acc = bias[c];
        for (int k = 0; k < KS; k++) acc += P[((int64_t)k * S + r) * d + c];
        a = a + acc;                       // x + (attn/mlp output), mod.rs:346-348
        x[(int64_t)r * d + c] = a;
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
    if (c < d) h[(int64_t)r * d + c] = (v[i] - mean) / denom * g[c] + b[c];
  }
}

// ---- skinny GEMM: P[ks][r][n] = sum_{k in slice ks} in[r][k] * W[k][n],  r < n_rows <= S ----
// block = 4 waves, column tile 128 (32 lanes x float4), each half-wave owns 4 consecutive k per
// 32-deep iteration; input rows staged in LDS with the prologue applied.
constexpr int GV_CT = 128;
constexpr int GV_KSL_MAX = 512;

template <int MR>
__global__ __launch_bounds__(256) void dec_gemv_kernel(GemvArgs a) {
  __shared__ __attribute__((aligned(16))) float xs[MR * GV_KSL_MAX];   // also the cross-wave reduction buffer
  static_assert(MR * GV_KSL_MAX >= 4 * MR * GV_CT, "reduction buffer must fit");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, c4 = (lane & 31) * 4;
  const int n0 = blockIdx.x * GV_CT, ks = blockIdx.y;
  const int k0 = ks * a.KSL;
  const int kn = min(a.KSL, a.K - k0);          // rows of this slice (multiple of 32)
  const int n_rows = a.st[ST_N];
  const int unit = wave * 2 + half;
  const bool col_ok = (n0 + c4) < a.ldw;
  const float* Wp = a.W + (int64_t)k0 * a.ldw + n0 + c4;

  for (int r0 = 0; r0 < n_rows; r0 += MR) {
    // ---- stage in[r0..r0+MR)[k0..k0+kn) with the prologue ----
    for (int e = tid; e < MR * kn; e += 256) {
      const int r = e / kn, kk = e - r * kn;
      const int row = r0 + r, k = k0 + kk;
      float v = 0.f;
      if (row < n_rows) {
        if (a.pro == PRO_PLAIN) {
          v = a.src[(int64_t)row * a.ld_src + k];
        } else if (a.pro == PRO_GELU) {           // GELU(lin1(x)), mod.rs:377-378, lin1 partials folded here
          float acc = a.pbias[k];
          for (int s = 0; s < a.KSp; s++) acc += a.src[((int64_t)s * a.S + row) * a.ld_src + k];
          v = gelu_erf(acc);
        } else {                                  // PRO_ATTN: combine the key-chunk partials of cross-attention
          const int hh = k >> 6, dh = k & 63;
          const float* ca = a.src + ((int64_t)(row * a.n_head + hh) * a.n_chunks) * CA_STRIDE;
          float M = -1.0e30f;
          for (int c = 0; c < a.n_chunks; c++) M = fmaxf(M, ca[c * CA_STRIDE]);
          float num = 0.f, den = 0.f;
          for (int c = 0; c < a.n_chunks; c++) {
            const float w = expf(ca[c * CA_STRIDE] - M);
            num += w * ca[c * CA_STRIDE + 2 + dh];
            den += w * ca[c * CA_STRIDE + 1];
          }
          v = num / den;
        }
      }
      xs[r * GV_KSL_MAX + kk] = v;
    }
    __syncthreads();
    float acc[MR][4];
#pragma unroll
    for (int r = 0; r < MR; r++) { acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.f; }
    if (col_ok) {
      for (int kb = unit * 4; kb < kn; kb += 32) {
        const float* wp = Wp + (int64_t)kb * a.ldw;
        const float4 w0 = *reinterpret_cast<const float4*>(wp);
        const float4 w1 = *reinterpret_cast<const float4*>(wp + a.ldw);
        const float4 w2 = *reinterpret_cast<const float4*>(wp + 2 * (int64_t)a.ldw);
        const float4 w3 = *reinterpret_cast<const float4*>(wp + 3 * (int64_t)a.ldw);
#pragma unroll
        for (int r = 0; r < MR; r++) {
          const float4 xv = *reinterpret_cast<const float4*>(&xs[r * GV_KSL_MAX + kb]);
          acc[r][0] += xv.x * w0.x; acc[r][1] += xv.x * w0.y; acc[r][2] += xv.x * w0.z; acc[r][3] += xv.x * w0.w;
          acc[r][0] += xv.y * w1.x; acc[r][1] += xv.y * w1.y; acc[r][2] += xv.y * w1.z; acc[r][3] += xv.y * w1.w;
          acc[r][0] += xv.z * w2.x; acc[r][1] += xv.z * w2.y; acc[r][2] += xv.z * w2.z; acc[r][3] += xv.z * w2.w;
          acc[r][0] += xv.w * w3.x; acc[r][1] += xv.w * w3.y; acc[r][2] += xv.w * w3.z; acc[r][3] += xv.w * w3.w;
        }
      }
    }
    __syncthreads();   // everyone is done reading xs: reuse it as red[4][MR][128]
#pragma unroll
    for (int r = 0; r < MR; r++)
#pragma unroll
      for (int c = 0; c < 4; c++) acc[r][c] += __shfl_xor(acc[r][c], 32);
    if (half == 0) {
#pragma unroll
      for (int r = 0; r < MR; r++)
        *reinterpret_cast<float4*>(&xs[(wave * MR + r) * GV_CT + c4]) =
            make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
    }
    __syncthreads();
    for (int e = tid; e < MR * GV_CT; e += 256) {
      const int r = e / GV_CT, c = e - r * GV_CT;
      const int row = r0 + r, col = n0 + c;
      if (row < n_rows && col < a.N) {
        const float v = (xs[(0 * MR + r) * GV_CT + c] + xs[(1 * MR + r) * GV_CT + c]) +
                        (xs[(2 * MR + r) * GV_CT + c] + xs[(3 * MR + r) * GV_CT + c]);
        a.P[((int64_t)ks * a.S + row) * a.N + col] = v;
      }
    }
    __syncthreads();
  }
}

// ---- masked self-attention over the paged self-KV cache, one wave per (beam, head) ------
constexpr int SA_MAXPOS = 448;
__global__ __launch_bounds__(64) void dec_self_attn_kernel(const int* __restrict__ st, StepLayout lay,
                                                           const float* __restrict__ Pqkv, int KS,
                                                           const float* __restrict__ bqkv, int d,
                                                           float* __restrict__ Kc, float* __restrict__ Vc,
                                                           const int* __restrict__ tab, int Lmax, float scale,
                                                           float* __restrict__ att) {
  __shared__ __attribute__((aligned(16))) float qs[64];
  __shared__ __attribute__((aligned(16))) float knew[64];
  __shared__ float vnew[64];
  __shared__ float ps[SA_MAXPOS];
  const int i = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
  if (i >= st[ST_N]) return;
  const int len = st[lay.len + i];
  const int col = h * 64 + lane;
  // fold the QKV partials: q = (xWq + bq) * s, k = (xWk) * s, v = xWv + bv  (mod.rs:429-431, :506-514)
  float q = bqkv[col], k = bqkv[d + col], v = bqkv[2 * d + col];
  for (int s = 0; s < KS; s++) {
    const float* p = Pqkv + ((int64_t)s * lay.S + i) * (3 * d);
    q += p[col]; k += p[d + col]; v += p[2 * d + col];
  }
  q *= scale; k *= scale;
  const int* tb = tab + i * Lmax;
  const int newrow = tb[len - 1];
  Kc[(int64_t)newrow * d + col] = k;
  Vc[(int64_t)newrow * d + col] = v;
  qs[lane] = q; knew[lane] = k; vnew[lane] = v;
  __syncthreads();
  constexpr int NPL = SA_MAXPOS / 64;
  float sc[NPL];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < NPL; j++) {
    const int p = lane + j * 64;
    sc[j] = -INFINITY;
    if (p < len) {
      float acc = 0.f;
      if (p == len - 1) {
#pragma unroll
        for (int c = 0; c < 64; c++) acc += qs[c] * knew[c];
      } else {
        const float4* kr = reinterpret_cast<const float4*>(Kc + (int64_t)tb[p] * d + h * 64);
#pragma unroll
        for (int c = 0; c < 16; c++) {
          const float4 kv = kr[c];
          const float4 qv = *reinterpret_cast<const float4*>(&qs[4 * c]);
          acc += qv.x * kv.x + qv.y * kv.y + qv.z * kv.z + qv.w * kv.w;
        }
      }
      sc[j] = acc;
      m = fmaxf(m, acc);
    }
  }
  m = wave_max(m);
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < NPL; j++) {
    const int p = lane + j * 64;
    if (p < len) {
      const float e = expf(sc[j] - m);
      ps[p] = e;
      l += e;
    }
  }
  l = wave_sum(l);
  __syncthreads();
  float o = 0.f;
  for (int p = 0; p < len - 1; p++) o += ps[p] * Vc[(int64_t)tb[p] * d + col];
  o += ps[len - 1] * vnew[lane];
  att[(int64_t)i * d + col] = o / l;
}

// ---- cross-attention over one key chunk of one window's cached K/V, all of its beams -----
constexpr int CA_CH = 128;   // keys per chunk
__global__ __launch_bounds__(256) void dec_cross_attn_kernel(const int* __restrict__ st, StepLayout lay,
                                                             const float* __restrict__ Pq, int KS,
                                                             const float* __restrict__ bq, int d,
                                                             const float* __restrict__ ckv, int ldkv, int koff,
                                                             const int* __restrict__ win_row0,
                                                             const int* __restrict__ win_C, float scale,
                                                             int n_head, int n_chunks, float* __restrict__ ca) {
  __shared__ __attribute__((aligned(16))) float Kt[CA_CH][65];
  __shared__ __attribute__((aligned(16))) float qs[MAX_BEAMS][64];
  __shared__ float sp[2][MAX_BEAMS][CA_CH];
  __shared__ float pb[MAX_BEAMS][CA_CH];
  __shared__ float stat[MAX_BEAMS][2];
  __shared__ float ored[4][MAX_BEAMS][64];
  const int c = blockIdx.x, h = blockIdx.y, w = blockIdx.z, tid = threadIdx.x;
  const int nb = st[lay.win_nb + w];
  if (nb == 0) return;
  const int* slots = st + lay.win_slots + w * MAX_BEAMS;
  const int C = win_C[w];
  const int j0 = c * CA_CH;
  const int nk = min(CA_CH, C - j0);
  if (nk <= 0) {   // chunk past this window's encoder length: neutral partial
    for (int e = tid; e < nb * CA_STRIDE; e += 256) {
      const int b = e / CA_STRIDE, f = e - b * CA_STRIDE;
      ca[((int64_t)(slots[b] * n_head + h) * n_chunks + c) * CA_STRIDE + f] = (f == 0) ? -1.0e30f : 0.f;
    }
    return;
  }
  // q = (x Wq + bq) * s  (mod.rs:483, :506-509)
  for (int e = tid; e < nb * 64; e += 256) {
    const int b = e >> 6, dh = e & 63, col = h * 64 + dh;
    float q = bq[col];
    for (int s = 0; s < KS; s++) q += Pq[((int64_t)s * lay.S + slots[b]) * d + col];
    qs[b][dh] = q * scale;
  }
  const float* Kb = ckv + (int64_t)(win_row0[w] + j0) * ldkv + koff + h * 64;   // K pre-scaled at projection time
  const float* Vb = Kb + d;
  for (int e = tid; e < CA_CH * 16; e += 256) {
    const int r = e >> 4, q4 = (e & 15) * 4;
    float4 kv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < nk) kv = *reinterpret_cast<const float4*>(Kb + (int64_t)r * ldkv + q4);
    Kt[r][q4 + 0] = kv.x; Kt[r][q4 + 1] = kv.y; Kt[r][q4 + 2] = kv.z; Kt[r][q4 + 3] = kv.w;
  }
  __syncthreads();
  {
    const int j = tid & (CA_CH - 1), hf = tid >> 7;
    float acc[MAX_BEAMS];
#pragma unroll
    for (int b = 0; b < MAX_BEAMS; b++) acc[b] = 0.f;
#pragma unroll 8
    for (int dh = hf * 32; dh < hf * 32 + 32; dh++) {
      const float kv = Kt[j][dh];
#pragma unroll
      for (int b = 0; b < MAX_BEAMS; b++) acc[b] += qs[b][dh] * kv;   // rows b >= nb hold stale q: never read back
    }
#pragma unroll
    for (int b = 0; b < MAX_BEAMS; b++) sp[hf][b][j] = acc[b];
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
    const int dh = tid & 63, gq = tid >> 6;
    float o[MAX_BEAMS];
#pragma unroll
    for (int b = 0; b < MAX_BEAMS; b++) o[b] = 0.f;
    const int jb = gq * 32, je = min(jb + 32, nk);
    for (int j = jb; j < je; j++) {
      const float vv = Vb[(int64_t)j * ldkv + dh];
#pragma unroll
      for (int b = 0; b < MAX_BEAMS; b++) o[b] += pb[b][j] * vv;
    }
#pragma unroll
    for (int b = 0; b < MAX_BEAMS; b++) ored[gq][b][dh] = o[b];
  }
  __syncthreads();
  for (int e = tid; e < nb * 64; e += 256) {
    const int b = e >> 6, dh = e & 63;
    float* dst = ca + ((int64_t)(slots[b] * n_head + h) * n_chunks + c) * CA_STRIDE;
    dst[2 + dh] = (ored[0][b][dh] + ored[1][b][dh]) + (ored[2][b][dh] + ored[3][b][dh]);
    if (dh == 0) { dst[0] = stat[b][0]; dst[1] = stat[b][1]; }
  }
}

// ---- mask + log_softmax + top-k of one beam's logits row (transcribe.rs:271-304) ---------
struct Cand { float v; int id; };
__device__ __forceinline__ bool better(float v, int id, float bv, int bid) {
  return v > bv || (v == bv && id < bid);
}

__global__ __launch_bounds__(1024) void dec_topk_kernel(const int* __restrict__ st, const float* __restrict__ logits,
                                                         int KS, int64_t plane, int V,
                                                         const float* __restrict__ mask, int use_mask,
                                                         int k, int32_t* __restrict__ out_id,
                                                         float* __restrict__ out_lp, float* __restrict__ row_stats) {
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
  float m = -INFINITY;
  for (int c = tid; c < V; c += 1024) {
    float v = x[c];
    for (int s2 = 1; s2 < KS; s2++) v += x[(int64_t)s2 * plane + c];   // K-split logits partials, fixed order
    if (use_mask) v += mask[c];
    m = fmaxf(m, v);
    // sorted insertion, (value desc, id asc); ids arrive ascending so equal values never displace
    if (v > tv[TOPK_MAX - 1]) {
      float cv = v; int ci = c;
#pragma unroll
      for (int j = 0; j < TOPK_MAX; j++) {
        if (cv > tv[j]) { float t = tv[j]; int u = ti[j]; tv[j] = cv; ti[j] = ci; cv = t; ci = u; }
      }
    }
  }
  m = wave_max(m);
  if (lane == 0) redv[wave] = m;
  __syncthreads();
  if (tid == 0) {
    float mm = redv[0];
    for (int j = 1; j < 16; j++) mm = fmaxf(mm, redv[j]);
    bc[0] = mm;
  }
  __syncthreads();
  const float M = bc[0];
  float s = 0.f;
  for (int c = tid; c < V; c += 1024) {
    float v = x[c];
    for (int s2 = 1; s2 < KS; s2++) v += x[(int64_t)s2 * plane + c];
    if (use_mask) v += mask[c];
    s += expf(v - M);
  }
  s = wave_sum(s);
  __syncthreads();
  if (lane == 0) redv[wave] = s;
  __syncthreads();
  if (tid == 0) {
    float ss = 0.f;
    for (int j = 0; j < 16; j++) ss += redv[j];
    bc[1] = logf(ss);
    row_stats[2 * r] = M; row_stats[2 * r + 1] = bc[1];
  }
  __syncthreads();
  const float lse = bc[1];
  // k rounds of block-wide argmax over the threads' list heads
  for (int round = 0; round < k; round++) {
    float bv = tv[0]; int bi = ti[0];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
      if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    __syncthreads();
    if (lane == 0) { redv[wave] = bv; redi[wave] = bi; }
    __syncthreads();
    float gv = redv[0]; int gi = redi[0];
    for (int j = 1; j < 16; j++)
      if (better(redv[j], redi[j], gv, gi)) { gv = redv[j]; gi = redi[j]; }
    if (tid == 0) {
      out_id[r * TOPK_MAX + round] = gi;
      out_lp[r * TOPK_MAX + round] = (gv - M) - lse;   // log_softmax, transcribe.rs:276
    }
    if (ti[0] == gi) {   // the winner pops its head
#pragma unroll
      for (int j = 0; j < TOPK_MAX - 1; j++) { tv[j] = tv[j + 1]; ti[j] = ti[j + 1]; }
      tv[TOPK_MAX - 1] = -INFINITY; ti[TOPK_MAX - 1] = 0x7fffffff;
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

void launch_dec_prepare(hipStream_t st, const int* state, const StepLayout& lay, int n_max, const int* tab_old,
                        int* tab_new, int Lmax, const float* E, const float* pos, int d, float* x) {
  hipLaunchKernelGGL(dec_prepare_kernel, dim3(n_max), dim3(128), 0, st, state, lay, tab_old, tab_new, Lmax, E, pos, d,
                     x);
}

void launch_dec_resolve_ln(hipStream_t st, const int* state, int n_max, float* x, const float* P, int KS, int S,
                           const float* bias, int d, const LayerNormW& ln, int eps_inside_sqrt, float* h) {
  if (d <= 1024)
    hipLaunchKernelGGL(dec_resolve_ln_kernel<4>, dim3(n_max), dim3(256), 0, st, state, x, P, KS, S, bias, d, ln.g,
                       ln.b, ln.eps, eps_inside_sqrt, h);
  else
    hipLaunchKernelGGL(dec_resolve_ln_kernel<8>, dim3(n_max), dim3(256), 0, st, state, x, P, KS, S, bias, d, ln.g,
                       ln.b, ln.eps, eps_inside_sqrt, h);
}

void gemv_plan(int K, int N, int* KS, int* KSL) {
  // slices of >= 64 rows, <= GV_KSL_MAX, at most 16 partials, aiming at >= ~512 blocks
  const int tiles = (N + GV_CT - 1) / GV_CT;
  int ks = std::max(1, std::min(16, 512 / std::max(tiles, 1)));
  ks = std::min(ks, std::max(1, K / 64));
  int ksl = ((K + ks - 1) / ks + 31) / 32 * 32;
  if (ksl > GV_KSL_MAX) { ksl = GV_KSL_MAX; }
  ks = (K + ksl - 1) / ksl;
  *KS = ks; *KSL = ksl;
}

void launch_dec_gemv(hipStream_t st, const GemvArgs& a, int n_rows_hint) {
  dim3 grid((a.N + GV_CT - 1) / GV_CT, a.KS);
  if (n_rows_hint <= 4)
    hipLaunchKernelGGL(dec_gemv_kernel<4>, grid, dim3(256), 0, st, a);
  else if (n_rows_hint <= 8)
    hipLaunchKernelGGL(dec_gemv_kernel<8>, grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(dec_gemv_kernel<16>, grid, dim3(256), 0, st, a);
}

void launch_dec_self_attn(hipStream_t st, const int* state, const StepLayout& lay, int n_max, int n_head,
                          const float* Pqkv, int KS, const float* bqkv, int d, float* Kc, float* Vc, const int* tab,
                          int Lmax, float scale, float* att) {
  hipLaunchKernelGGL(dec_self_attn_kernel, dim3(n_max, n_head), dim3(64), 0, st, state, lay, Pqkv, KS, bqkv, d, Kc, Vc,
                     tab, Lmax, scale, att);
}

void launch_dec_cross_attn(hipStream_t st, const int* state, const StepLayout& lay, int n_windows, int n_head,
                           int n_chunks, const float* Pq, int KS, const float* bq, int d, const float* ckv, int ldkv,
                           int koff, const int* win_row0, const int* win_C, float scale, float* ca) {
  hipLaunchKernelGGL(dec_cross_attn_kernel, dim3(n_chunks, n_head, n_windows), dim3(256), 0, st, state, lay, Pq, KS, bq,
                     d, ckv, ldkv, koff, win_row0, win_C, scale, n_head, n_chunks, ca);
}

void launch_dec_topk(hipStream_t st, const int* state, int n_max, const float* logits, int KS, int64_t plane, int V,
                     const float* mask, int use_mask, int k, int32_t* out_id, float* out_lp, float* row_stats) {
  hipLaunchKernelGGL(dec_topk_kernel, dim3(n_max), dim3(1024), 0, st, state, logits, KS, plane, V, mask, use_mask, k,
                     out_id, out_lp, row_stats);
}

void launch_dec_logprob_row(hipStream_t st, const float* x, int KS, int64_t plane, int V, const float* mask,
                            int use_mask, const float* stats, float* out) {
  hipLaunchKernelGGL(dec_logprob_row_kernel, dim3((V + 255) / 256), dim3(256), 0, st, x, KS, plane, V, mask, use_mask,
                     stats, out);
}

int cross_attn_chunk() { return CA_CH; }

}  // namespace wb
