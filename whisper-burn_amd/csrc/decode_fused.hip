// Fused small-batch decode-step kernels: one launch per sublayer instead of one per matrix.
//
// With a handful of live beams a decode step is bound by its chain of DEPENDENT launches, not by bytes
// (tiny.en: 30 launches of 4-6 us around 17 us of streaming).  Two producer-aligned fusions remove kernel
// boundaries without any grid-wide synchronisation, because the block that OWNS a slice of an intermediate
// can apply the next matrix to it on its own and leave a partial plane that the consumer's prologue already
// knows how to fold (fixed order, no atomics -> bit-reproducible):
//
//   dec_mlp_fused_kernel    block j owns hidden units [64 j, 64 j + 64):  LN(x + pending) -> W1[:, slice]
//                           -> + b1 -> GELU -> W2[slice, :]  -> plane j of [4 d / 64][S][d]   (mod.rs:376-382)
//   dec_attn_fused_kernel   block h owns head h:  LN(x + pending) -> Wqkv[:, head h] -> cache append -> masked
//                           self-attention over the paged cache -> Wo[head h rows, :] -> plane h of [H][S][d]
//                                                                            (mod.rs:428-436, :493-533)
//
// Every block redoes the row prologue (fold + LayerNorm: a few KB from L2); each weight is still read once.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "decode.h"
#include "decode_fused_bodies.h"
#include "wave_ops.h"

namespace wb {
namespace {

using namespace fused;

// one launch per sublayer: grid = the roles of that sublayer, the step state comes from a.st
template <int MR, int DPL, bool REC>
__global__ __launch_bounds__(512) void dec_mlp_fused_kernel(MlpFusedArgs a) {
  if (blockIdx.y > 0) {                            // row group g: rows [g MR, g MR + MR) (9 - 16 live rows; plain planes only)
    const int r0 = blockIdx.y * MR;
    const int64_t sh = (int64_t)r0 * a.d;
    a.row0 = r0; a.x_in += sh; a.x_out += sh; a.P += sh;
    if (a.pend) a.pend += sh;
  }
  dec_mlp_body<MR, DPL, REC, false>(a, blockIdx.x, PsStep());
}
// grid = (8, rows): workgroups go round-robin over the 8 XCDs by linear id, so x = head puts every beam's block of
// one head on the SAME XCD -- the head's weight slices cross the fabric once and are shared through that L2
template <int DPL>
__global__ __launch_bounds__(512) void dec_attn_fused_kernel(AttnFusedArgs a) {
  dec_attn_body<DPL, false>(a, blockIdx.x, blockIdx.y, PsStep());
}
template <int DPL, int NP>
__global__ __launch_bounds__(512) void dec_cross_fused_kernel(CrossFusedArgs a) {
  dec_cross_body<DPL, false, NP>(a, blockIdx.x, blockIdx.y, PsStep());
}


// ---- the MLP sublayer for 9 - 16 live rows on the matrix cores (round 6) ------------------------------------------------
// Beam search over a 30 s chunk keeps 15 rows live (3 windows x 5 beams).  dec_mlp_fused_kernel serves them as two row groups
// of 8: every block streams its W1 / W2 slices twice and spends 16 x 2 x d x 64 FMAs on the vector pipe -- 13.7 us per layer,
// the longest kernel of a beam step (profiles/r06_p_kernel_stats_tiny_en_30s.csv).  Same partition here (block j owns hidden
// units [64 j, 64 j + 64), plane j of [4 d / 64][S][d] out, lin2 bias left to the consumer), ONE pass over 16 rows, both
// products on v_mfma_f32_16x16x4_f32 (exact f32) with the skinny kernel's operand scheme: rows k-major in LDS (A operand of lane
// l for K-rows k .. k + 3 = word 16 k + l), one float4 per lane = 4 K-rows x 64 columns of the weight per load, component c =
// the B operand of accumulator c (columns 4 j + c).  Phase 1: the 8 waves split K = d, partial 16 x 64 tiles meet in LDS in
// wave order; + b1, GELU (mod.rs:377-378).  Phase 2: wave w owns output columns [64 w, 64 w + 64) over the whole K = 64.
typedef float mm_f32x4 __attribute__((ext_vector_type(4)));
template <int DPL>
__global__ __launch_bounds__(512) void dec_mlp16_mfma_kernel(MlpFusedArgs a) {
  constexpr int MR = 16, d = 64 * DPL, HS = 64, NT = 512;
  constexpr int KW = d / 8;                        // K-rows per wave in phase 1 (16 / 48 / 64)
  constexpr int NL1 = KW / 4, NL2 = HS / 4;        // float4 loads per lane: phase 1 (4 / 12 / 16), phase 2 (16)
  __shared__ __attribute__((aligned(16))) float xT[d * MR];           // LN(x + pending), [k][16]
  __shared__ __attribute__((aligned(16))) float red[8][MR][HS];       // phase-1 partial tiles, one per wave
  __shared__ __attribute__((aligned(16))) float hT[HS * MR];          // GELU(hidden slice), [k][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jblk = blockIdx.x, j0 = jblk * HS;
  const int krow = lane >> 4, cq = lane & 15;
  const int n_rows = min(MR, a.st[ST_N]);
  // the weights first: nothing they need depends on the rows
  float4 w1[NL1], w2[NL2];
  {
    const float* p1 = a.W1 + (int64_t)(wave * KW + krow) * a.ld1 + j0 + 4 * cq;
#pragma unroll
    for (int t = 0; t < NL1; t++) w1[t] = *reinterpret_cast<const float4*>(p1 + (int64_t)(4 * t) * a.ld1);
    const int strip = wave < d / 64 ? wave : 0;    // (waves past the last column strip re-read strip 0: unused)
    const float* p2 = a.W2 + (int64_t)(j0 + krow) * d + strip * 64 + 4 * cq;
#pragma unroll
    for (int t = 0; t < NL2; t++) w2[t] = *reinterpret_cast<const float4*>(p2 + (int64_t)(4 * t) * d);
  }
  if (n_rows <= 0) return;
  // ---- x + (bias + pending planes), s ascending (mod.rs:346-348); LayerNorm; staged k-major.  One wave per row (two rows each).
#pragma unroll 1
  for (int r = wave; r < MR; r += 8) {
    if (r >= n_rows) {
      for (int c = lane; c < d; c += 64) xT[c * MR + r] = 0.f;
      continue;
    }
    float gv[DPL], bv[DPL], v[DPL];
    const float* xr = a.x_in + (int64_t)r * d;
#pragma unroll
    for (int i = 0; i < DPL; i++) { gv[i] = a.ln_g[lane + 64 * i]; bv[i] = a.ln_b[lane + 64 * i]; v[i] = xr[lane + 64 * i]; }
    if (a.KSp > 0) {
      float acc[DPL];
#pragma unroll
      for (int i = 0; i < DPL; i++) acc[i] = a.pbias[lane + 64 * i];
      const float* pp = a.pend + (int64_t)r * d;
      const int64_t plane = (int64_t)a.S * d;
      constexpr int CH = 8;
      for (int sp = 0; sp < a.KSp; sp += CH) {
        float t[CH][DPL];
#pragma unroll
        for (int j = 0; j < CH; j++) {
          const float* pj = pp + (int64_t)min(sp + j, a.KSp - 1) * plane;
#pragma unroll
          for (int i = 0; i < DPL; i++) t[j][i] = pj[lane + 64 * i];
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
    for (int i = 0; i < DPL; i++) { sm += v[i]; if (jblk == 0) a.x_out[(int64_t)r * d + lane + 64 * i] = v[i]; }
    const float mean = wave_sum(sm) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < DPL; i++) { const float t = v[i] - mean; q += t * t; }
    const float var = wave_sum(q) / (float)d;
    const float denom = a.ln_inside ? sqrtf(var + a.ln_eps) : (sqrtf(var) + a.ln_eps);
#pragma unroll
    for (int i = 0; i < DPL; i++) xT[(lane + 64 * i) * MR + r] = (v[i] - mean) / denom * gv[i] + bv[i];
  }
  __syncthreads();
  // ---- phase 1: hidden[16][64] partials over this wave's K range
  {
    mm_f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; c++) acc[c] = mm_f32x4{0.f, 0.f, 0.f, 0.f};
    const float* ap = xT + (wave * KW) * MR + lane;
#pragma unroll
    for (int t = 0; t < NL1; t++) {
      const float av = ap[(4 * t) * MR];
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w1[t].x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w1[t].y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w1[t].z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w1[t].w, acc[3], 0, 0, 0);
    }
    // accumulator c, register i of lane l: row 4 (l / 16) + i, column 4 (l % 16) + c
#pragma unroll
    for (int i = 0; i < 4; i++)
      *reinterpret_cast<float4*>(&red[wave][4 * krow + i][4 * cq]) = make_float4(acc[0][i], acc[1][i], acc[2][i], acc[3][i]);
  }
  __syncthreads();
  // ---- + b1, GELU (exact erf), k-major for phase 2: 16 x 64 values over 512 threads
#pragma unroll
  for (int e = tid; e < MR * HS; e += NT) {
    const int r = e / HS, c = e - r * HS;
    float hsum = a.b1[j0 + c];
#pragma unroll
    for (int w8 = 0; w8 < 8; w8++) hsum += red[w8][r][c];                 // wave order fixed
    hT[c * MR + r] = r < n_rows ? gelu_erf_f(hsum) : 0.f;
  }
  __syncthreads();
  // ---- phase 2: plane j, columns [64 wave, 64 wave + 64), K = the 64 hidden units of the slice
  if (wave < d / 64) {
    mm_f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; c++) acc[c] = mm_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NL2; t++) {
      const float av = hT[(4 * t) * MR + lane];
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w2[t].x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w2[t].y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w2[t].z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w2[t].w, acc[3], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = 4 * krow + i;
      if (r < n_rows)
        *reinterpret_cast<float4*>(&a.P[((int64_t)jblk * a.S + r) * d + wave * 64 + 4 * cq]) =
            make_float4(acc[0][i], acc[1][i], acc[2][i], acc[3][i]);
    }
  }
}

}  // namespace

bool dec_fused_supported(int d) { return d == 128 || d == 384 || d == 512; }
int dec_mlp_fused_planes(int d) { return 4 * d / 64; }

void launch_dec_mlp_fused(hipStream_t st, const MlpFusedArgs& a, int n_rows_hint) {
  // 9 - 16 rows with plain planes: one pass on the matrix cores (WHISPER_HIP_MLP16_MFMA=0: two row groups of 8 on the vector pipe)
  static const bool mfma16 = []() { const char* e = getenv("WHISPER_HIP_MLP16_MFMA"); return !(e && e[0] == '0'); }();
  if (mfma16 && n_rows_hint > 8 && n_rows_hint <= 16 && a.n_chunks == 0 && a.row0 == 0) {
    const dim3 grid16(4 * a.d / 64), block16(512);
    if (a.d == 128) WB_KLAUNCH((dec_mlp16_mfma_kernel<2>), grid16, block16, 0, st, a);
    else if (a.d == 384) WB_KLAUNCH((dec_mlp16_mfma_kernel<6>), grid16, block16, 0, st, a);
    else WB_KLAUNCH((dec_mlp16_mfma_kernel<8>), grid16, block16, 0, st, a);
    return;
  }
  // more than 8 rows: row groups of 8 in grid.y (the caller guarantees plain planes: n_chunks == 0)
  const dim3 grid(4 * a.d / 64, n_rows_hint > 8 ? (n_rows_hint + 7) / 8 : 1), block(512);
#define WB_MLP(MR_, DPL_)                                                                          \
  do {                                                                                             \
    if (a.n_chunks > 0) WB_KLAUNCH((dec_mlp_fused_kernel<MR_, DPL_, true>), grid, block, 0, st, a); \
    else WB_KLAUNCH((dec_mlp_fused_kernel<MR_, DPL_, false>), grid, block, 0, st, a);               \
  } while (0)
  if (n_rows_hint <= 4) {
    if (a.d == 128) WB_MLP(4, 2); else if (a.d == 384) WB_MLP(4, 6); else WB_MLP(4, 8);
  } else {
    if (a.d == 128) WB_MLP(8, 2); else if (a.d == 384) WB_MLP(8, 6); else WB_MLP(8, 8);
  }
#undef WB_MLP
}

void launch_dec_cross_fused(hipStream_t st, const CrossFusedArgs& a, int n_rows_hint) {
  const dim3 grid(a.n_head <= 8 ? 8 : a.n_head, n_rows_hint), block(512);
  // (a.n_pass = 2: some window holds more than CROSS_FUSED_MAX_C keys -- the two-pass ring)
  if (a.n_pass <= 1) {
    if (a.d == 128) WB_KLAUNCH((dec_cross_fused_kernel<2, 1>), grid, block, 0, st, a);
    else if (a.d == 384) WB_KLAUNCH((dec_cross_fused_kernel<6, 1>), grid, block, 0, st, a);
    else WB_KLAUNCH((dec_cross_fused_kernel<8, 1>), grid, block, 0, st, a);
  } else {
    if (a.d == 128) WB_KLAUNCH((dec_cross_fused_kernel<2, 2>), grid, block, 0, st, a);
    else if (a.d == 384) WB_KLAUNCH((dec_cross_fused_kernel<6, 2>), grid, block, 0, st, a);
    else WB_KLAUNCH((dec_cross_fused_kernel<8, 2>), grid, block, 0, st, a);
  }
}

void launch_dec_attn_fused(hipStream_t st, const AttnFusedArgs& a, int n_rows_hint) {
  const dim3 grid(a.n_head <= 8 ? 8 : a.n_head, n_rows_hint), block(512);
  if (a.d == 128) WB_KLAUNCH((dec_attn_fused_kernel<2>), grid, block, 0, st, a);
  else if (a.d == 384) WB_KLAUNCH((dec_attn_fused_kernel<6>), grid, block, 0, st, a);
  else WB_KLAUNCH((dec_attn_fused_kernel<8>), grid, block, 0, st, a);
}

}  // namespace wb
