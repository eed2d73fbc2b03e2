// K2/K4: exact-f32 MFMA GEMM with fused epilogues, plus implicit-GEMM A loaders for the conv stem.
//
//   C[M,N] = (act(A[M,K] * B[K,N] + bias[N]) * col_scale) + residual[M,N] + aux[aux_idx[m], N]
//
// Replaces Burn's nn::Linear (y = x W + b, W stored [d_in, d_out]; used at
// /root/reference/src/model/mod.rs:377-379, :429-435, :483-489), the tied-embedding logits
// matmul (mod.rs:156) and -- through the A loaders -- conv::Conv1d k=3 p=1 s=1/2
// (mod.rs:243-244) with GELU, the [B,d,C]->[B,C,d] transpose (mod.rs:246) and the
// positional add (mod.rs:247-252) fused into the epilogue.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- f32 in, f32 accumulate, bit-identical to a k-ordered
// fmaf chain (157 TF peak on MI355X).  Tiles are LDS-staged k-major ([BK][BM], [BK][BN]) so
// every MFMA operand is one conflict-free ds_read_b32; the global->LDS path is register
// double-buffered (one barrier per 16-deep k-tile).
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace wb {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NT = 256;
enum { AMODE_ROWS = 0, AMODE_CONV1 = 1 };

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// PF = register prefetch depth in k-tiles.  1: the next tile loads while the current one multiplies (the big
// encoder GEMMs: the grid alone keeps HBM busy).  3: the split-K decode GEMMs -- their grids are a few hundred
// blocks of 3-10 tiles, so bytes in flight per CU, not tile latency, bound them: each block keeps three tiles
// (60 KB) in flight.
template <int BM, int BN, int WGM, int WGN, int AMODE, int BK, int PF = 1>
__global__ __launch_bounds__(NT) void gemm_f32_kernel(GemmArgs g) {
  constexpr int KQ = BK / 4;   // float4 quads per A row per k-tile
  constexpr int TM = BM / WGM, TN = BN / WGN;
  constexpr int RM = TM / 32, RN = TN / 32;
  constexpr int LDA_S = BM + 4, LDB_S = BN + 4;
  static_assert(WGM * WGN == 4 && TM % 32 == 0 && TN % 32 == 0, "bad tiling");
  __shared__ __attribute__((aligned(16))) float As[2][BK][LDA_S];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB_S];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int li = lane & 31, lh = lane >> 5;
  // XCD-aware block order is not needed here: neighbouring blocks share B panels through L2 either way
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int M = g.M, N = g.N, K = g.K;
  // split-K (decode batch mode): slice blockIdx.z of the K range, raw partial sums to C + z * c_split_stride
  const int kchunk = g.ksplit > 1 ? ((K + g.ksplit - 1) / g.ksplit + 31) / 32 * 32 : K;   // launchers keep K-slices multiples of 32
  const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
  float* Cout = g.C + (int64_t)blockIdx.z * g.c_split_stride;

  // ---- A staging: per-thread rows are fixed over the k loop ----
  constexpr int A_F4 = (AMODE == AMODE_ROWS) ? (BM * KQ + NT - 1) / NT : 0;   // float4 loads / thread / tile
  constexpr int A_EL = (AMODE == AMODE_CONV1) ? (BM * BK) / NT : 0;          // scalar loads / thread / tile
  const float* a_row[A_F4 > 0 ? A_F4 : 1];
  int a_klo[A_F4 > 0 ? A_F4 : 1], a_khi[A_F4 > 0 ? A_F4 : 1];
  int c1_off[A_EL > 0 ? A_EL : 1], c1_flag[A_EL > 0 ? A_EL : 1];
  if constexpr (AMODE == AMODE_ROWS) {
#pragma unroll
    for (int i = 0; i < A_F4; i++) {
      int idx = tid + i * NT, r = idx / KQ;
      int m = m0 + r;
      a_row[i] = nullptr; a_klo[i] = 0; a_khi[i] = 0;
      if (r < BM && m < M) {
        if (g.a_desc) {
          RowDesc d = g.a_desc[m];
          a_row[i] = g.A + d.off; a_klo[i] = d.klo; a_khi[i] = d.khi;
        } else {
          a_row[i] = g.A + (int64_t)m * g.lda; a_klo[i] = 0; a_khi[i] = K;
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < A_EL; i++) {
      int idx = tid + i * NT, r = idx % BM;
      int m = m0 + r;
      c1_off[i] = 0; c1_flag[i] = 4;   // 4 = row out of range
      if (m < M) {
        RowDesc d = g.a_desc[m];
        c1_off[i] = d.off; c1_flag[i] = (d.klo ? 1 : 0) | (d.khi ? 2 : 0);
      }
    }
  }
  constexpr int B_F4 = (BK * BN / 4) / NT;   // float4 loads / thread / tile
  static_assert((BK * BN / 4) % NT == 0, "B tile must divide over the block");

  float4 ra_[PF][A_F4 > 0 ? A_F4 : 1];
  float rc1_[PF][A_EL > 0 ? A_EL : 1];
  float4 rb_[PF][B_F4];

  auto load_tile = [&](int k0, auto& ra, auto& rc1, auto& rb) {
    if constexpr (AMODE == AMODE_ROWS) {
#pragma unroll
      for (int i = 0; i < A_F4; i++) {
        int idx = tid + i * NT, kq = (idx % KQ) * 4;
        int k = k0 + kq;
        // the load only: masking of a partially valid quad happens in store_tile, when the registers are consumed --
        // touching the value here would put an s_waitcnt vmcnt(0) behind every prefetch
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a_row[i] != nullptr && k + 3 >= a_klo[i] && k < a_khi[i]) v = *reinterpret_cast<const float4*>(a_row[i] + k);
        ra[i] = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_EL; i++) {
        int idx = tid + i * NT, kk_t = idx / BM;
        int k = k0 + kk_t;
        int ci = k / 3, kk = k - ci * 3;
        bool zero = (c1_flag[i] & 4) || (kk == 0 && (c1_flag[i] & 1)) || (kk == 2 && (c1_flag[i] & 2));
        rc1[i] = zero ? 0.f : g.A[(int64_t)c1_off[i] + (int64_t)ci * g.conv1_tstride + kk - 1];
      }
    }
#pragma unroll
    for (int i = 0; i < B_F4; i++) {
      int idx = tid + i * NT, kr = idx / (BN / 4), n4 = (idx % (BN / 4)) * 4;
      int n = n0 + n4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < g.ldb) v = *reinterpret_cast<const float4*>(g.B + (int64_t)(k0 + kr) * g.ldb + n);
      rb[i] = v;
    }
  };
  auto store_tile = [&](int buf, int k0, auto& ra, auto& rc1, auto& rb) {
    if constexpr (AMODE == AMODE_ROWS) {
#pragma unroll
      for (int i = 0; i < A_F4; i++) {
        int idx = tid + i * NT, r = idx / KQ, kq = (idx % KQ) * 4;
        if (r < BM) {
          const int k = k0 + kq;
          if (k < a_klo[i] || k + 3 >= a_khi[i]) {   // partially masked quad (not hit by aligned callers)
            if (k + 0 < a_klo[i] || k + 0 >= a_khi[i]) ra[i].x = 0.f;
            if (k + 1 < a_klo[i] || k + 1 >= a_khi[i]) ra[i].y = 0.f;
            if (k + 2 < a_klo[i] || k + 2 >= a_khi[i]) ra[i].z = 0.f;
            if (k + 3 < a_klo[i] || k + 3 >= a_khi[i]) ra[i].w = 0.f;
          }
          As[buf][kq + 0][r] = ra[i].x; As[buf][kq + 1][r] = ra[i].y;
          As[buf][kq + 2][r] = ra[i].z; As[buf][kq + 3][r] = ra[i].w;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_EL; i++) {
        int idx = tid + i * NT;
        As[buf][idx / BM][idx % BM] = rc1[i];
      }
    }
#pragma unroll
    for (int i = 0; i < B_F4; i++) {
      int idx = tid + i * NT, kr = idx / (BN / 4), n4 = (idx % (BN / 4)) * 4;
      *reinterpret_cast<float4*>(&Bs[buf][kr][n4]) = rb[i];
    }
  };

  f32x16 acc[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; i++)
#pragma unroll
    for (int j = 0; j < RN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const int nk = max(0, kend - kbeg) / BK;
  // register slot of tile t: t % PF.  Tiles 0 .. PF-1 are requested up front; tile t + PF is requested as
  // soon as tile t has left its slot for LDS.
#pragma unroll
  for (int u = 0; u < PF; u++)
    if (u < nk) load_tile(kbeg + u * BK, ra_[u], rc1_[u], rb_[u]);
  if (nk > 0) {
    store_tile(0, kbeg, ra_[0], rc1_[0], rb_[0]);
    if (PF < nk) load_tile(kbeg + PF * BK, ra_[0], rc1_[0], rb_[0]);
  }
  __syncthreads();
  for (int t0 = 0; t0 < nk; t0 += PF) {
#pragma unroll
   for (int u = 0; u < PF; u++) {
    const int t = t0 + u;
    if (t >= nk) break;
    const int buf = t & 1;
#pragma unroll
    for (int kk = 0; kk < BK / 2; kk++) {
      const int kidx = 2 * kk + lh;
      float a[RM], b[RN];
#pragma unroll
      for (int i = 0; i < RM; i++) a[i] = As[buf][kidx][wm * TM + i * 32 + li];
#pragma unroll
      for (int j = 0; j < RN; j++) b[j] = Bs[buf][kidx][wn * TN + j * 32 + li];
#pragma unroll
      for (int i = 0; i < RM; i++)
#pragma unroll
        for (int j = 0; j < RN; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nk) {
      store_tile(buf ^ 1, kbeg + (t + 1) * BK, ra_[(u + 1) % PF], rc1_[(u + 1) % PF], rb_[(u + 1) % PF]);
      if (t + 1 + PF < nk) load_tile(kbeg + (t + 1 + PF) * BK, ra_[(u + 1) % PF], rc1_[(u + 1) % PF], rb_[(u + 1) % PF]);
    }
    __syncthreads();
   }
  }

  // ---- epilogue ----
  // Straight-line per 32x32 sub-tile: all residual / positional reads of its 16 rows are issued back to back from
  // clamped addresses (a per-row branch would serialise them into 16 dependent round trips), then the arithmetic,
  // then predicated stores.
#pragma unroll
  for (int i = 0; i < RM; i++)
#pragma unroll
    for (int j = 0; j < RN; j++) {
      const int col = n0 + wn * TN + j * 32 + li;
      const bool col_ok = col < N;
      const int colc = col_ok ? col : N - 1;
      // (column blocks: GemmArgs::c_block_cols)
      const int64_t cbase = g.c_block_cols > 0 ? (int64_t)(col / g.c_block_cols) * g.c_block_stride + col % g.c_block_cols : col;
      const float bias = g.bias ? g.bias[colc] : 0.f;
      float cs = 1.f;
      if (g.col_scale_period > 0 && (colc % g.col_scale_period) < g.col_scale_width) cs = g.col_scale;
      const int rbase = m0 + wm * TM + i * 32 + 4 * lh;
      float res[16], ax[16];
      if (g.residual) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = min(rbase + (r & 3) + 8 * (r >> 2), M - 1);
          res[r] = g.residual[(int64_t)row * g.ldr + colc];
        }
      }
      if (g.aux) {
        int ai[16];
#pragma unroll
        for (int r = 0; r < 16; r++) ai[r] = g.aux_idx[min(rbase + (r & 3) + 8 * (r >> 2), M - 1)];
#pragma unroll
        for (int r = 0; r < 16; r++) ax[r] = g.aux[(int64_t)ai[r] * g.ld_aux + colc];
      }
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        float v = acc[i][j][r] + bias;
        if (g.act == ACT_GELU) v = gelu_erf(v);
        if (g.col_scale_period > 0) v *= cs;
        if (g.residual) v = res[r] + v;
        if (g.aux) v = v + ax[r];
        if (col_ok && row < M) Cout[cbase + (int64_t)row * g.ldc] = v;
      }
    }
}

template <int BM, int BN, int WGM, int WGN, int AMODE, int BK = 16, int PF = 1>
void launch_cfg(hipStream_t st, const GemmArgs& a) {
  dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, a.ksplit > 1 ? a.ksplit : 1);
  WB_KLAUNCH((gemm_f32_kernel<BM, BN, WGM, WGN, AMODE, BK, PF>), grid, dim3(NT), 0, st, a);
}

#ifdef WB_GEMM_PROBE
// tools/gemm_probe.cpp: one tile / k-depth / prefetch configuration by number
int launch_gemm_f32_cfg(hipStream_t st, const GemmArgs& a, int cfg) {
  switch (cfg) {
    case 0: launch_cfg<64, 64, 2, 2, AMODE_ROWS, 16, 1>(st, a); break;
    case 1: launch_cfg<64, 64, 2, 2, AMODE_ROWS, 16, 2>(st, a); break;
    case 2: launch_cfg<64, 64, 2, 2, AMODE_ROWS, 16, 4>(st, a); break;
    case 3: launch_cfg<64, 64, 2, 2, AMODE_ROWS, 32, 1>(st, a); break;
    case 4: launch_cfg<64, 64, 2, 2, AMODE_ROWS, 32, 2>(st, a); break;
    case 5: launch_cfg<64, 64, 2, 2, AMODE_ROWS, 32, 3>(st, a); break;
    case 6: launch_cfg<128, 128, 2, 2, AMODE_ROWS, 16, 1>(st, a); break;
    case 7: launch_cfg<128, 128, 2, 2, AMODE_ROWS, 16, 2>(st, a); break;
    case 8: launch_cfg<128, 128, 2, 2, AMODE_ROWS, 16, 3>(st, a); break;
    case 9: launch_cfg<128, 64, 2, 2, AMODE_ROWS, 16, 2>(st, a); break;
    case 10: launch_cfg<128, 64, 2, 2, AMODE_ROWS, 32, 2>(st, a); break;
    case 11: launch_cfg<64, 128, 2, 2, AMODE_ROWS, 32, 2>(st, a); break;
    case 12: launch_cfg<32, 128, 1, 4, AMODE_ROWS, 32, 2>(st, a); break;
    case 13: launch_cfg<32, 128, 1, 4, AMODE_ROWS, 32, 3>(st, a); break;
    default: return -1;
  }
  return 0;
}
#endif

}  // namespace

int launch_gemm_f32(hipStream_t st, const GemmArgs& a) {
  if (a.M <= 0 || a.N <= 0) return 0;
  if (a.K % 16 != 0 || a.ldb % 4 != 0) return -1;
  if (a.ksplit > 1 && (a.bias || a.residual || a.aux || a.act != ACT_NONE || a.col_scale_period > 0)) return -1;
  const bool conv1 = a.conv1_tstride > 0;
  // pick the largest tile that still gives the 256 CUs >= ~1.5 waves of blocks
  auto blocks = [&](int bm, int bn) { return (int64_t)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); };
  if (conv1) {
    if (blocks(128, 128) >= 384) launch_cfg<128, 128, 2, 2, AMODE_CONV1>(st, a);
    else launch_cfg<64, 64, 2, 2, AMODE_CONV1>(st, a);
    return 0;
  }
  if (a.ksplit > 1 && a.K % 32 != 0) return -1;
  // Measured on MI355X with tools/gemm_probe.cpp (bit-identical results across configurations: the k order of
  // the accumulation chain never changes).  Many-block shapes (>= 3 blocks per CU at 128x128) are fastest on
  // 128x128 tiles with two k-tiles in flight; everything smaller -- the bench's 3-window encoder, the N = d
  // projections of `small` -- on 32x128 tiles (1x4 waves), 32-deep k-tiles, two in flight.
  if (a.ksplit > 1) {
    if (a.M <= 32) launch_cfg<32, 128, 1, 4, AMODE_ROWS, 32, 3>(st, a);
    else launch_cfg<64, 64, 2, 2, AMODE_ROWS, 32, 3>(st, a);
  } else if (a.K % 32 != 0) {
    if (a.M <= 32) launch_cfg<32, 128, 1, 4, AMODE_ROWS>(st, a);
    else launch_cfg<64, 64, 2, 2, AMODE_ROWS>(st, a);
  } else if (blocks(128, 128) >= 768) {
    launch_cfg<128, 128, 2, 2, AMODE_ROWS, 16, 2>(st, a);
  } else {
    launch_cfg<32, 128, 1, 4, AMODE_ROWS, 32, 2>(st, a);
  }
  return 0;
}

}  // namespace wb
