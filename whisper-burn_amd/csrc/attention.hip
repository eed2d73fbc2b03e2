// K5: flash-style attention in exact f32 on MFMA, head size 64.
//
// Replaces qkv_attention (/root/reference/src/model/mod.rs:493-533), which materialises the
// [B,H,Lq,Lk] score tensor through >= 9 launches: scores never leave registers here.
//
// "Swapped" formulation so that softmax rows are lane-local and P never round-trips LDS:
//   S^T[kv][q]  = K_tile * Q^T          (A = K from LDS, B = Q fragment held in registers)
//   O^T[dh][q] += V_tile^T * P^T        (A = V from LDS, B = P^T = the S^T accumulator itself)
// With v_mfma_f32_32x32x2_f32, the accumulator register r of lane (j, h) holds element
// (row = (r&3) + 8(r>>2) + 4h, col = j); as the B operand of the PV product the same lane must
// supply P^T[kv(step, h)][q = j] -- exactly its own accumulator r = step when the V operand is
// gathered with the matching kv permutation.  Row max / sum are per lane (+ one xor-32 exchange).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "kernels.h"
#include "wave_ops.h"

namespace wb {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KV_TILE = 32;
constexpr int KS_LD = 65;   // padded K rows: lanes walk rows -> conflict-free ds_read_b32
constexpr int VS_LD = 64;

template <int NW>
__global__ __launch_bounds__(NW * 64) void attention_f32_kernel(const float* __restrict__ Q, int ldq,
                                                                  const float* __restrict__ K,
                                                                  const float* __restrict__ V, int ldkv,
                                                                  float* __restrict__ O, int ldo,
                                                                  const AttnSeg* __restrict__ segs, float scale,
                                                                  int causal) {
  __shared__ __attribute__((aligned(16))) float Ks[2][KV_TILE][KS_LD];
  __shared__ __attribute__((aligned(16))) float Vs[2][KV_TILE][VS_LD];
  constexpr int NTHR = NW * 64;
  constexpr int LD_PER_T = (KV_TILE * 16) / NTHR;   // float4 loads per thread per operand per tile
  static_assert((KV_TILE * 16) % NTHR == 0, "tile must divide over the block");

  const AttnSeg seg = segs[blockIdx.z];
  const int head = blockIdx.y;
  const int q_base = blockIdx.x * 32 * NW;
  if (q_base >= seg.q_len) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int q0 = q_base + wave * 32;
  const int qi = q0 + li;                         // this lane's query position (may be >= q_len)
  const int qrow = min(qi, seg.q_len - 1);

  // Q fragment: Q[q][head*64 + hh*32 + s], s = 0..31, pre-scaled (mod.rs:506-509)
  float qreg[32];
  {
    const float4* qp = reinterpret_cast<const float4*>(Q + (int64_t)(seg.q_row0 + qrow) * ldq + head * 64 + hh * 32);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      float4 v = qp[i];
      qreg[4 * i + 0] = v.x * scale; qreg[4 * i + 1] = v.y * scale;
      qreg[4 * i + 2] = v.z * scale; qreg[4 * i + 3] = v.w * scale;
    }
  }

  int kv_end = seg.kv_len;
  if (causal) kv_end = min(kv_end, q_base + 32 * NW);   // keys beyond the block's last query are masked
  const int n_tiles = (kv_end + KV_TILE - 1) / KV_TILE;

  const float* Kb = K + (int64_t)seg.kv_row0 * ldkv + head * 64;
  const float* Vb = V + (int64_t)seg.kv_row0 * ldkv + head * 64;
  float4 rk[LD_PER_T], rv[LD_PER_T];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < LD_PER_T; i++) {
      int idx = tid + i * NTHR, r = idx >> 4, c4 = (idx & 15) * 4;
      int kv = t * KV_TILE + r;
      if (kv < seg.kv_len) {
        rk[i] = *reinterpret_cast<const float4*>(Kb + (int64_t)kv * ldkv + c4);
        rv[i] = *reinterpret_cast<const float4*>(Vb + (int64_t)kv * ldkv + c4);
      } else {
        rk[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        rv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LD_PER_T; i++) {
      int idx = tid + i * NTHR, r = idx >> 4, c4 = (idx & 15) * 4;
      Ks[buf][r][c4 + 0] = rk[i].x * scale; Ks[buf][r][c4 + 1] = rk[i].y * scale;   // mod.rs:510-514
      Ks[buf][r][c4 + 2] = rk[i].z * scale; Ks[buf][r][c4 + 3] = rk[i].w * scale;
      *reinterpret_cast<float4*>(&Vs[buf][r][c4]) = rv[i];
    }
  };

  f32x16 oacc[2];
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) oacc[t][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;

  if (n_tiles > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();
  for (int t = 0; t < n_tiles; t++) {
    const int buf = t & 1;
    if (t + 1 < n_tiles) load_tile(t + 1);
    // ---- S^T = K_tile * Q^T ----
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; r++) sacc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 32; s++)
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[buf][li][hh * 32 + s], qreg[s], sacc, 0, 0, 0);
    // ---- mask + online softmax (rows = this lane's query) ----
    const int kv0 = t * KV_TILE + 4 * hh;
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int kv = kv0 + (r & 3) + 8 * (r >> 2);
      const bool ok = kv < seg.kv_len && (!causal || kv <= qi);
      sacc[r] = ok ? sacc[r] : -INFINITY;
      tmax = fmaxf(tmax, sacc[r]);
    }
    tmax = xor32_max(tmax);
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      sacc[r] = expf(sacc[r] - m_new);
      psum += sacc[r];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    // ---- O^T = alpha O^T + V_tile^T * P^T: the tile's product is summed on its own (a chain of 32 keys) and joins the running
    // sum with one fma per element -- a single chain over all keys (750 per window) is ~4x less accurate
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
      f32x16 pt;
#pragma unroll
      for (int r = 0; r < 16; r++) pt[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 16; s++)
        pt = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[buf][(s & 3) + 8 * (s >> 2) + 4 * hh][32 * tt + li], sacc[s], pt, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; r++) oacc[tt][r] = fmaf(oacc[tt][r], alpha, pt[r]);
    }
    if (t + 1 < n_tiles) store_tile(buf ^ 1);
    __syncthreads();
  }

  const float l_tot = xor32_sum(l_run);
  if (qi < seg.q_len) {
    float* op = O + (int64_t)(seg.q_row0 + qi) * ldo + head * 64;
#pragma unroll
    for (int tt = 0; tt < 2; tt++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        float4 v = make_float4(oacc[tt][4 * g + 0] / l_tot, oacc[tt][4 * g + 1] / l_tot,
                               oacc[tt][4 * g + 2] / l_tot, oacc[tt][4 * g + 3] / l_tot);
        *reinterpret_cast<float4*>(op + 32 * tt + 8 * g + 4 * hh) = v;
      }
  }
}

// The same arithmetic for SMALL grids (a few windows: the 128-query blocks above leave most of the 256 CUs idle and
// every wave walks the whole key range alone).  One block per 32 queries; its NW waves split the KEY tiles
// (wave w takes tiles w, w+NW, ...), so nothing is shared between waves: K and V go straight from global memory
// into the MFMA operand layout (a lane's K operand is 32 contiguous floats of one key row, its V operands are
// 128-byte row segments) -- no LDS, no barrier in the loop.  The NW partial (max, sum, O) triples are merged once
// through LDS in a fixed order (deterministic; differs from the single-chain result only by f32 rounding).
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void attention_f32_kvsplit_kernel(const float* __restrict__ Q, int ldq,
                                                                          const float* __restrict__ K,
                                                                          const float* __restrict__ V, int ldkv,
                                                                          float* __restrict__ O, int ldo,
                                                                          const AttnSeg* __restrict__ segs, float scale,
                                                                          int causal) {
  __shared__ float part[NW][34][64];
  const AttnSeg seg = segs[blockIdx.z];
  const int head = blockIdx.y;
  const int q_base = blockIdx.x * 32;
  if (q_base >= seg.q_len) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int qi = q_base + li;
  const int qrow = min(qi, seg.q_len - 1);

  float qreg[32];
  {
    const float4* qp = reinterpret_cast<const float4*>(Q + (int64_t)(seg.q_row0 + qrow) * ldq + head * 64 + hh * 32);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      float4 v = qp[i];
      qreg[4 * i + 0] = v.x * scale; qreg[4 * i + 1] = v.y * scale;
      qreg[4 * i + 2] = v.z * scale; qreg[4 * i + 3] = v.w * scale;
    }
  }
  int kv_end = seg.kv_len;
  if (causal) kv_end = min(kv_end, q_base + 32);
  const int n_tiles = (kv_end + KV_TILE - 1) / KV_TILE;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);                     // wave-uniform on the scalar side
  const float* Kh = K + (int64_t)seg.kv_row0 * ldkv + head * 64;              // uniform bases; lane parts are 32-bit offsets
  const float* Vh = V + (int64_t)seg.kv_row0 * ldkv + head * 64;
  const int koff_lane = li * ldkv + hh * 32;                                   // this lane's half of key row li
  const int voff_lane = 4 * hh * ldkv + li;                                    // this lane's column, rows 4hh + perm(s)

  auto load_tile = [&](int t, float4 (&kr)[8], float (&vr)[32]) {
    const bool full = (t + 1) * KV_TILE <= seg.kv_len;
    const float* kt = Kh + (int64_t)t * KV_TILE * ldkv;
    const float* vt = Vh + (int64_t)t * KV_TILE * ldkv;
    if (full) {
      const float4* kp = reinterpret_cast<const float4*>(kt + koff_lane);
#pragma unroll
      for (int i = 0; i < 8; i++) kr[i] = kp[i];
#pragma unroll
      for (int s = 0; s < 16; s++) {
        const float* vp = vt + ((s & 3) + 8 * (s >> 2)) * ldkv;              // uniform row base
        vr[s] = vp[voff_lane];
        vr[16 + s] = vp[voff_lane + 32];
      }
    } else {                                                                   // the last tile of a segment: rows past the end are zeros
      const bool kok = t * KV_TILE + li < seg.kv_len;
#pragma unroll
      for (int i = 0; i < 8; i++) kr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kok) {
        const float4* kp = reinterpret_cast<const float4*>(kt + koff_lane);
#pragma unroll
        for (int i = 0; i < 8; i++) kr[i] = kp[i];
      }
#pragma unroll
      for (int s = 0; s < 16; s++) {
        const int r = t * KV_TILE + 4 * hh + (s & 3) + 8 * (s >> 2);
        const float* vp = vt + ((s & 3) + 8 * (s >> 2)) * ldkv;
        vr[s] = 0.f; vr[16 + s] = 0.f;                                         // P is exactly 0 there, but 0 * garbage must stay 0
        if (r < seg.kv_len) { vr[s] = vp[voff_lane]; vr[16 + s] = vp[voff_lane + 32]; }
      }
    }
  };

  f32x16 oacc[2];
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) oacc[t][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;

  auto compute = [&](int t, float4 (&kr)[8], float (&vr)[32]) {
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; r++) sacc[r] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[i].x * scale, qreg[4 * i + 0], sacc, 0, 0, 0);   // mod.rs:510-514
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[i].y * scale, qreg[4 * i + 1], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[i].z * scale, qreg[4 * i + 2], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[i].w * scale, qreg[4 * i + 3], sacc, 0, 0, 0);
    }
    const int kv0 = t * KV_TILE + 4 * hh;
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int kv = kv0 + (r & 3) + 8 * (r >> 2);
      const bool ok = kv < seg.kv_len && (!causal || kv <= qi);
      sacc[r] = ok ? sacc[r] : -INFINITY;
      tmax = fmaxf(tmax, sacc[r]);
    }
    tmax = xor32_max(tmax);
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      sacc[r] = expf(sacc[r] - m_new);
      psum += sacc[r];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    // (one chain per wave: the waves split the key tiles, so a chain covers a quarter of the keys and the four partial sums
    // meet below -- already a two-level sum; a separate tile accumulator as in the kernel above would not fit 256 registers)
#pragma unroll
    for (int tt = 0; tt < 2; tt++)
#pragma unroll
      for (int r = 0; r < 16; r++) oacc[tt][r] *= alpha;
#pragma unroll
    for (int tt = 0; tt < 2; tt++)
#pragma unroll
      for (int s = 0; s < 16; s++)
        oacc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[16 * tt + s], sacc[s], oacc[tt], 0, 0, 0);
  };

  {
    float4 k0[8], k1[8];
    float v0[32], v1[32];
    int t = wave_u;
    if (t < n_tiles) load_tile(t, k0, v0);
    while (t < n_tiles) {
      if (t + NW < n_tiles) load_tile(t + NW, k1, v1);
      compute(t, k0, v0);
      t += NW;
      if (t >= n_tiles) break;
      if (t + NW < n_tiles) load_tile(t + NW, k0, v0);
      compute(t, k1, v1);
      t += NW;
    }
  }

  // ---- merge the NW partial results (fixed order w = 0 .. NW-1) ----
#pragma unroll
  for (int tt = 0; tt < 2; tt++)
#pragma unroll
    for (int r = 0; r < 16; r++) part[wave][16 * tt + r][lane] = oacc[tt][r];
  part[wave][32][lane] = m_run;
  part[wave][33][lane] = l_run;
  __syncthreads();
  float m_all = -1.0e30f;
#pragma unroll
  for (int w = 0; w < NW; w++) m_all = fmaxf(m_all, part[w][32][lane]);
  float sc[NW], l_all = 0.f;
#pragma unroll
  for (int w = 0; w < NW; w++) {
    sc[w] = expf(part[w][32][lane] - m_all);
    l_all += part[w][33][lane] * sc[w];
  }
  const float l_tot = xor32_sum(l_all);
  if (qi < seg.q_len) {
    // wave w finishes accumulator registers [32/NW * w, 32/NW * (w+1)): whole float4 groups of the output row
    constexpr int GPW = 8 / NW;                 // float4 groups per wave
    static_assert(8 % NW == 0, "NW must divide the 8 output groups");
    float* op = O + (int64_t)(seg.q_row0 + qi) * ldo + head * 64;
#pragma unroll
    for (int gi = 0; gi < GPW; gi++) {
      const int gg = wave * GPW + gi, tt = gg >> 2, g = gg & 3;
      float o[4];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w++) a += part[w][16 * tt + 4 * g + c][lane] * sc[w];
        o[c] = a / l_tot;
      }
      *reinterpret_cast<float4*>(op + 32 * tt + 8 * g + 4 * hh) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}


// ---- K5 on the 16-bit matrix path (round 6) ----------------------------------------------------------------------------------
// The same two products on v_mfma_f32_32x32x16_f16 with SPLIT-PRECISION operands (gemm_f16x3.hip's scheme: x = hi + 2^-11 lo,
// both fp16; a product of two fp16 values is exact in f32, accumulation is f32):
//   S^T = K_hi Q_hi^T + 2^-11 (K_hi Q_lo^T + K_lo Q_hi^T)             scores feed an exp: kept f32-grade (22-bit operands)
//   O^T += V_hi^T P_hi^T + 2^-11 (V_hi^T P_lo^T + V_lo^T P_hi^T)       P = exp(s - m) in (0, 1], split like everything else
// 24 MFMAs of 32 cycles per 32 x 32 key tile and wave instead of 64 of 64 cycles; softmax statistics stay f32 and lane-local
// (the swapped formulation of the f32 kernel above).  Operand layouts: the A operand of lane (li, hh) in k-step ks is the 8
// consecutive halves [16 ks + 8 hh, + 8) of row li -- one 16-byte LDS read from K_hi / K_lo [key][64 + pad]; the B operand
// of the PV product is the lane's OWN score registers: MFMA k-slot (step s, half hh, j) is bound to key
// (j & 3) + 8 (2 s + (j >> 2)) + 4 hh of the tile -- the key of accumulator register r = 8 s + j -- and V^T sits in LDS
// [dh][32 + pad] with its keys in that slot order, so its A operand is one 16-byte read as well.
// Range: |q s|, |k s|, |v| < 65504 or the affected outputs are inf / NaN -- which the range guard of the split-precision GEMM
// that consumes them (the out-projection) reports; the engine picks this kernel only while that GEMM is in use (engine.cpp).
typedef _Float16 a_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short a_u16;
constexpr int KH_LD = 64 + 8;    // halves per K row (144 B)
constexpr int VT_LD = 32 + 8;    // halves per V^T row (80 B)
constexpr float A_LO_SCALE = 2048.f, A_LO_UNSCALE = 1.f / 2048.f;

__device__ __forceinline__ void a_split(float x, _Float16& hi, _Float16& lo) {
  hi = (_Float16)x;
  lo = (_Float16)((x - (float)hi) * A_LO_SCALE);
}
__device__ __forceinline__ a_u16 a_bits(_Float16 h) { return __builtin_bit_cast(a_u16, h); }
// slot position of tile key kv (0 .. 31) in a V^T row: see the header comment
__device__ __forceinline__ int a_vslot(int kv) { return 16 * (kv >> 4) + 8 * ((kv >> 2) & 1) + (kv & 3) + 4 * ((kv >> 3) & 1); }

template <int NW>
__global__ __launch_bounds__(NW * 64) void attention_f16x3_kernel(const float* __restrict__ Q, int ldq,
                                                                    const float* __restrict__ K,
                                                                    const float* __restrict__ V, int ldkv,
                                                                    float* __restrict__ O, int ldo,
                                                                    const AttnSeg* __restrict__ segs, float scale,
                                                                    int causal, a_u16* __restrict__ Oh,
                                                                    a_u16* __restrict__ Ol) {
  __shared__ __attribute__((aligned(16))) a_u16 Khs[2][KV_TILE][KH_LD];
  __shared__ __attribute__((aligned(16))) a_u16 Kls[2][KV_TILE][KH_LD];
  __shared__ __attribute__((aligned(16))) a_u16 Vth[2][64][VT_LD];
  __shared__ __attribute__((aligned(16))) a_u16 Vtl[2][64][VT_LD];
  constexpr int NTHR = NW * 64;
  constexpr int LD_PER_T = (KV_TILE * 16) / NTHR;
  static_assert((KV_TILE * 16) % NTHR == 0, "tile must divide over the block");

  const AttnSeg seg = segs[blockIdx.z];
  const int head = blockIdx.y;
  const int q_base = blockIdx.x * 32 * NW;
  if (q_base >= seg.q_len) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hh = lane >> 5;
  const int q0 = q_base + wave * 32;
  const int qi = q0 + li;
  const int qrow = min(qi, seg.q_len - 1);

  // Q fragments: k-step ks holds head dims [16 ks + 8 hh, + 8) of this lane's query, pre-scaled (mod.rs:506-509), split
  a_f16x8 qh[4], ql[4];
  {
    const float* qp = Q + (int64_t)(seg.q_row0 + qrow) * ldq + head * 64 + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const float4 a = *reinterpret_cast<const float4*>(qp + 16 * ks), b = *reinterpret_cast<const float4*>(qp + 16 * ks + 4);
      const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
      for (int i = 0; i < 8; i++) { _Float16 h, l; a_split(v[i] * scale, h, l); qh[ks][i] = h; ql[ks][i] = l; }
    }
  }

  int kv_end = seg.kv_len;
  if (causal) kv_end = min(kv_end, q_base + 32 * NW);
  const int n_tiles = (kv_end + KV_TILE - 1) / KV_TILE;

  const float* Kb = K + (int64_t)seg.kv_row0 * ldkv + head * 64;
  const float* Vb = V + (int64_t)seg.kv_row0 * ldkv + head * 64;
  float4 rk[LD_PER_T], rv[LD_PER_T];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < LD_PER_T; i++) {
      const int idx = tid + i * NTHR, r = idx >> 4, c4 = (idx & 15) * 4;
      const int kv = t * KV_TILE + r;
      if (kv < seg.kv_len) {
        rk[i] = *reinterpret_cast<const float4*>(Kb + (int64_t)kv * ldkv + c4);
        rv[i] = *reinterpret_cast<const float4*>(Vb + (int64_t)kv * ldkv + c4);
      } else {
        rk[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        rv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LD_PER_T; i++) {
      const int idx = tid + i * NTHR, r = idx >> 4, c4 = (idx & 15) * 4;
      const float kv4[4] = {rk[i].x * scale, rk[i].y * scale, rk[i].z * scale, rk[i].w * scale};   // mod.rs:510-514
      const float vv4[4] = {rv[i].x, rv[i].y, rv[i].z, rv[i].w};
      a_u16 kh[4], kl[4];
      const int vs = a_vslot(r);
#pragma unroll
      for (int c = 0; c < 4; c++) {
        _Float16 h, l;
        a_split(kv4[c], h, l); kh[c] = a_bits(h); kl[c] = a_bits(l);
        a_split(vv4[c], h, l);
        Vth[buf][c4 + c][vs] = a_bits(h); Vtl[buf][c4 + c][vs] = a_bits(l);
      }
      *reinterpret_cast<uint2*>(&Khs[buf][r][c4]) = make_uint2((unsigned)kh[0] | ((unsigned)kh[1] << 16), (unsigned)kh[2] | ((unsigned)kh[3] << 16));
      *reinterpret_cast<uint2*>(&Kls[buf][r][c4]) = make_uint2((unsigned)kl[0] | ((unsigned)kl[1] << 16), (unsigned)kl[2] | ((unsigned)kl[3] << 16));
    }
  };

  f32x16 oacc[2];
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) oacc[t][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;

  if (n_tiles > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();
  for (int t = 0; t < n_tiles; t++) {
    const int buf = t & 1;
    if (t + 1 < n_tiles) load_tile(t + 1);
    // ---- S^T = K_tile * Q^T, three terms ----
    f32x16 sacc, sacl;
#pragma unroll
    for (int r = 0; r < 16; r++) { sacc[r] = 0.f; sacl[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const a_f16x8 ah = *reinterpret_cast<const a_f16x8*>(&Khs[buf][li][16 * ks + 8 * hh]);
      const a_f16x8 al = *reinterpret_cast<const a_f16x8*>(&Kls[buf][li][16 * ks + 8 * hh]);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[ks], sacc, 0, 0, 0);
      sacl = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[ks], sacl, 0, 0, 0);
      sacl = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[ks], sacl, 0, 0, 0);
    }
    // ---- mask + online softmax (rows = this lane's query) ----
    const int kv0 = t * KV_TILE + 4 * hh;
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int kv = kv0 + (r & 3) + 8 * (r >> 2);
      const bool ok = kv < seg.kv_len && (!causal || kv <= qi);
      sacc[r] = ok ? sacc[r] + sacl[r] * A_LO_UNSCALE : -INFINITY;
      tmax = fmaxf(tmax, sacc[r]);
    }
    tmax = xor32_max(tmax);
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = expf(m_run - m_new);
    float psum = 0.f;
    a_f16x8 ph[2], pl[2];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float pv = expf(sacc[r] - m_new);
      psum += pv;
      _Float16 h, l;
      a_split(pv, h, l);
      ph[r >> 3][r & 7] = h; pl[r >> 3][r & 7] = l;      // register r = 8 s + j is k-slot (s, hh, j) of the PV product
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    // ---- O^T = alpha O^T + V_tile^T * P^T (the tile's product summed on its own, as in the f32 kernel) ----
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
      f32x16 pt, ptl;
#pragma unroll
      for (int r = 0; r < 16; r++) { pt[r] = 0.f; ptl[r] = 0.f; }
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++) {
        const a_f16x8 vh = *reinterpret_cast<const a_f16x8*>(&Vth[buf][32 * tt + li][16 * s2 + 8 * hh]);
        const a_f16x8 vl = *reinterpret_cast<const a_f16x8*>(&Vtl[buf][32 * tt + li][16 * s2 + 8 * hh]);
        pt = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[s2], pt, 0, 0, 0);
        ptl = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[s2], ptl, 0, 0, 0);
        ptl = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[s2], ptl, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; r++) oacc[tt][r] = fmaf(oacc[tt][r], alpha, pt[r] + ptl[r] * A_LO_UNSCALE);
    }
    if (t + 1 < n_tiles) store_tile(buf ^ 1);
    __syncthreads();
  }

  const float l_tot = xor32_sum(l_run);
  if (qi < seg.q_len) {
    const int64_t orow = (int64_t)(seg.q_row0 + qi) * ldo + head * 64;
#pragma unroll
    for (int tt = 0; tt < 2; tt++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const float v[4] = {oacc[tt][4 * g + 0] / l_tot, oacc[tt][4 * g + 1] / l_tot,
                            oacc[tt][4 * g + 2] / l_tot, oacc[tt][4 * g + 3] / l_tot};
        const int64_t o = orow + 32 * tt + 8 * g + 4 * hh;
        if (Oh) {                                    // the out-projection is a split-precision GEMM: fp16 pieces (GemmArgs::Ah)
          unsigned hb[4], lb[4];
#pragma unroll
          for (int c = 0; c < 4; c++) { _Float16 h, l; a_split(v[c], h, l); hb[c] = a_bits(h); lb[c] = a_bits(l); }
          *reinterpret_cast<uint2*>(Oh + o) = make_uint2(hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16));
          *reinterpret_cast<uint2*>(Ol + o) = make_uint2(lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16));
        } else {
          *reinterpret_cast<float4*>(O + o) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
  }
}

}  // namespace

static bool attention_kvsplit_enabled() {
  static const bool on = [] { const char* e = getenv("WHISPER_HIP_ATTN_KVSPLIT"); return !(e && e[0] == '0'); }();
  return on;
}

// split: the 16-bit matrix path (attention_f16x3_kernel) for the LDS-tiled shapes; the key-split kernel for small grids and the
// 64-query blocks stay exact f32
bool launch_attention(hipStream_t st, const float* Q, int ldq, const float* K, const float* V, int ldkv,
                      float* O, int ldo, const AttnSeg* segs_dev, int n_segs, int max_q_len, int n_head,
                      float scale, int causal, bool split, uint16_t* Oh, uint16_t* Ol) {
  if (n_segs <= 0 || max_q_len <= 0) return false;
  static const bool f16_enabled = [] { const char* e = getenv("WHISPER_HIP_ATTN_F16"); return !(e && e[0] == '0'); }();
  const int64_t blocks128 = (int64_t)((max_q_len + 127) / 128) * n_head * n_segs;
  const bool kvsplit = !causal && max_q_len >= 256 && blocks128 < 384 && attention_kvsplit_enabled();
  if (split && f16_enabled && !kvsplit && max_q_len > 64) {
    dim3 grid((max_q_len + 127) / 128, n_head, n_segs);
    const bool pieces = Oh != nullptr && Ol != nullptr;
    hipLaunchKernelGGL((attention_f16x3_kernel<4>), grid, dim3(256), 0, st, Q, ldq, K, V, ldkv, O, ldo, segs_dev,
                       scale, causal, pieces ? Oh : nullptr, pieces ? Ol : nullptr);
    return pieces;
  }
  launch_attention_f32(st, Q, ldq, K, V, ldkv, O, ldo, segs_dev, n_segs, max_q_len, n_head, scale, causal);
  return false;
}

void launch_attention_f32(hipStream_t st, const float* Q, int ldq, const float* K, const float* V, int ldkv,
                          float* O, int ldo, const AttnSeg* segs_dev, int n_segs, int max_q_len, int n_head,
                          float scale, int causal) {
  if (n_segs <= 0 || max_q_len <= 0) return;
  const int64_t blocks128 = (int64_t)((max_q_len + 127) / 128) * n_head * n_segs;
  if (!causal && max_q_len >= 256 && blocks128 < 384 && attention_kvsplit_enabled()) {
    // a few windows: 128-query blocks cannot fill 256 CUs; split the keys over the waves instead
    dim3 grid((max_q_len + 31) / 32, n_head, n_segs);
    hipLaunchKernelGGL((attention_f32_kvsplit_kernel<4>), grid, dim3(256), 0, st, Q, ldq, K, V, ldkv, O, ldo, segs_dev,
                       scale, causal);
  } else if (max_q_len <= 64) {
    dim3 grid((max_q_len + 63) / 64, n_head, n_segs);
    hipLaunchKernelGGL((attention_f32_kernel<2>), grid, dim3(128), 0, st, Q, ldq, K, V, ldkv, O, ldo, segs_dev,
                       scale, causal);
  } else {
    dim3 grid((max_q_len + 127) / 128, n_head, n_segs);
    hipLaunchKernelGGL((attention_f32_kernel<4>), grid, dim3(256), 0, st, Q, ldq, K, V, ldkv, O, ldo, segs_dev,
                       scale, causal);
  }
}

}  // namespace wb
