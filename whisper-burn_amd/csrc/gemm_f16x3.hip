// Split-precision GEMM: exact-f32-grade products on the 16-bit matrix path (f32 accumulate), with the same fused
// epilogues as gemm.hip.
//
//   C[M,N] = epilogue( A[M,K] * W[K,N] ),   A = A_hi + 2^-11 A_lo,  W = W_hi + 2^-11 W_lo   (fp16 pieces)
//   A W  ~=  A_hi W_hi + 2^-11 (A_hi W_lo + A_lo W_hi)                                      (three MFMAs per pair)
//
// hi = fp16(x) and lo = fp16((x - hi) * 2^11) carry 22 of f32's 24 mantissa bits; a product of two fp16 values is exact in
// f32 and v_mfma_f32_32x32x16_f16 accumulates in f32, so the only terms lost are lo.lo (2^-22 relative) and the
// residual of the split itself.  Measured with the oracle as the model (tests/study_split_precision.py): encoder output
// and decoder log-probs as close to the f64 evaluation as plain f32 is (tiny.en 6.2e-5 vs 6.8e-5, small 4.6e-4 vs 3.3e-4;
// a bf16 split loses a decade, plain bf16 three).  Range: |x| < 65504 for every operand (encoder activations and weights
// stay below 10 on every fixture; the scaled low parts stay below the high parts' magnitude).  An operand outside that
// range turns its fp16 piece into inf and the affected outputs into inf / NaN: the epilogue raises GemmArgs::range_flag
// and the host repeats the pass with the exact-f32 kernel (engine.cpp: split_guarded); weights outside the range never
// get a split copy (model_load.cpp).
//
// The weight comes pre-split and K-contiguous ([N][K] fp16 x 2, made once at model load by split_weight_f16); the
// activations stay f32 in HBM and are split on their way into LDS.  16x the MFMA rate of the exact-f32 path for three
// times the instructions: the ceiling is 5.3x the f32 kernel's.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>
#include <utility>

#include "kernels.h"

namespace wb {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int NT = 256;
constexpr int BK = 32;
constexpr int LDS_LD = BK + 8;   // fp16 elements per LDS row
constexpr float LO_SCALE = 2048.f, LO_UNSCALE = 1.f / 2048.f;

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ u16 h_bits(_Float16 h) { return __builtin_bit_cast(u16, h); }
// x -> (hi, lo): hi = fp16(x) (round to nearest even), lo = fp16((x - hi) * 2^11)
__device__ __forceinline__ void split1(float x, u16& hi, u16& lo) {
  const _Float16 h = (_Float16)x;
  hi = h_bits(h);
  lo = h_bits((_Float16)((x - (float)h) * LO_SCALE));
}
__device__ __forceinline__ void split8(const float4& a, const float4& b, uint4& hi, uint4& lo) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  u16 h[8], l[8];
#pragma unroll
  for (int i = 0; i < 8; i++) split1(v[i], h[i], l[i]);
  hi = make_uint4((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16),
                  (unsigned)h[4] | ((unsigned)h[5] << 16), (unsigned)h[6] | ((unsigned)h[7] << 16));
  lo = make_uint4((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16),
                  (unsigned)l[4] | ((unsigned)l[5] << 16), (unsigned)l[6] | ((unsigned)l[7] << 16));
}

// Two waves per SIMD: bounded to 256 registers the 128 x 128 tile fits two blocks per CU (246 VGPRs, no scratch; unbounded the
// compiler took 198 + 128 accumulator registers = one block) -- large-v2 encoder 249.7 -> 196.5 ms on the same box
// (profiles/r03_r_*).
#ifndef WB_F16X3_MIN_WAVES
#define WB_F16X3_MIN_WAVES 2
#endif
#ifndef WB_F16X3_PF_SMALL
#define WB_F16X3_PF_SMALL 4      // k-tiles in flight of the small-tile configurations (see gemm_f16x3_kernel)
#endif
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
// PF: k-tiles requested ahead in registers.  The small-tile configurations serve the shapes with FEWER blocks than the chip has
// CUs (tiny.en's 3-window encoder: 174 blocks of 64 x 64): one block per CU, nothing to overlap a tile's memory round trip
// with, and a k-loop of 12 tiles cost 12 round trips (23 us per GEMM for 8 us of MFMA work).  They have registers to spare
// (110 of 256), so they keep PF = 4 tiles in flight; the 128 x 128 configuration (246 VGPRs, two blocks per CU) keeps PF = 1.
// APRE: the activations arrive as fp16 pieces (GemmArgs::Ah / Al, written once by their producer): an A item is two 16-byte
// loads that go to LDS as they are -- no split in the k-loop.
template <int BM, int BN, int WGM, int WGN, int PF, bool APRE>
__global__ __launch_bounds__(NT, WB_F16X3_MIN_WAVES) void gemm_f16x3_kernel(GemmArgs g, const u16* __restrict__ Wh,
                                                                            const u16* __restrict__ Wl, int ldwt) {
  constexpr int TM = BM / WGM, TN = BN / WGN;
  constexpr int RM = TM / 32, RN = TN / 32;
  static_assert(WGM * WGN == 4 && TM % 32 == 0 && TN % 32 == 0, "bad tiling");
  __shared__ __attribute__((aligned(16))) u16 Ah[2][BM][LDS_LD];
  __shared__ __attribute__((aligned(16))) u16 Al[2][BM][LDS_LD];
  __shared__ __attribute__((aligned(16))) u16 Bh[2][BN][LDS_LD];
  __shared__ __attribute__((aligned(16))) u16 Bl[2][BN][LDS_LD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int M = g.M, N = g.N, K = g.K;
  const int kchunk = g.ksplit > 1 ? ((K + g.ksplit - 1) / g.ksplit + BK - 1) / BK * BK : K;
  const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
  float* Cout = g.C + (int64_t)blockIdx.z * g.c_split_stride;

  // A: (row, k-octet) items, 8 f32 -> 8 + 8 fp16; B: (n, k-octet) items, 16 B of each piece
  constexpr int A_IT = (BM * (BK / 8) + NT - 1) / NT, B_IT = (BN * (BK / 8) + NT - 1) / NT;
  const float* a_row[A_IT];
  int64_t a_poff[A_IT];                            // APRE: element offset of the row in the piece planes (-1: no row)
  int a_klo[A_IT], a_khi[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; i++) {
    const int idx = tid + i * NT, r = idx / (BK / 8), m = m0 + r;
    a_row[i] = nullptr; a_klo[i] = 0; a_khi[i] = 0; a_poff[i] = -1;
    if (r < BM && m < M) {
      if constexpr (APRE) {
        a_poff[i] = (int64_t)m * g.lda;
      } else if (g.a_desc) {
        const RowDesc d = g.a_desc[m];
        a_row[i] = g.A + d.off; a_klo[i] = d.klo; a_khi[i] = d.khi;
      } else {
        a_row[i] = g.A + (int64_t)m * g.lda; a_khi[i] = K;
      }
    }
  }
  // A stays f32 in registers until store_tile: converting here would wait for the load right behind its issue
  float4 ra_lo[PF][A_IT], ra_hi[PF][A_IT];
  uint4 rbh[PF][B_IT], rbl[PF][B_IT];
  auto load_tile = [&](auto slot_c, int k0) {
    constexpr int sl = decltype(slot_c)::value;
#pragma unroll
    for (int i = 0; i < A_IT; i++) {
      const int idx = tid + i * NT, k = k0 + (idx % (BK / 8)) * 8;
      float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
      if constexpr (APRE) {
        // (ra_lo carries the octet's eight hi halves, ra_hi its eight lo halves: raw bits in the same registers)
        if (a_poff[i] >= 0) {
          lo = *reinterpret_cast<const float4*>(g.Ah + a_poff[i] + k);
          hi = *reinterpret_cast<const float4*>(g.Al + a_poff[i] + k);
        }
        ra_lo[sl][i] = lo; ra_hi[sl][i] = hi;
        continue;
      }
      // whole octets only: a row mask [klo, khi) must be a multiple of 8 at both ends -- the HOST guarantees it
      // (GemmArgs::a_mask_align, checked in gemm_dispatch; a per-element path for straddling octets cost the hot loop ~100
      // compare / branch instructions per k-tile for a case no caller has)
      if (a_row[i] != nullptr && k >= a_klo[i] && k + 7 < a_khi[i]) {
        lo = *reinterpret_cast<const float4*>(a_row[i] + k);
        hi = *reinterpret_cast<const float4*>(a_row[i] + k + 4);
      }
      ra_lo[sl][i] = lo; ra_hi[sl][i] = hi;
    }
#pragma unroll
    for (int i = 0; i < B_IT; i++) {
      const int idx = tid + i * NT, r = idx / (BK / 8), k = k0 + (idx % (BK / 8)) * 8, n = n0 + r;
      const bool ok = r < BN && n < N;
      rbh[sl][i] = ok ? *reinterpret_cast<const uint4*>(Wh + (int64_t)n * ldwt + k) : make_uint4(0, 0, 0, 0);
      rbl[sl][i] = ok ? *reinterpret_cast<const uint4*>(Wl + (int64_t)n * ldwt + k) : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_tile = [&](auto slot_c, int buf) {
    constexpr int sl = decltype(slot_c)::value;
#pragma unroll
    for (int i = 0; i < A_IT; i++) {
      const int idx = tid + i * NT, r = idx / (BK / 8), ko = (idx % (BK / 8)) * 8;
      if (r < BM) {
        uint4 hi, lo;
        if constexpr (APRE) {
          hi = __builtin_bit_cast(uint4, ra_lo[sl][i]); lo = __builtin_bit_cast(uint4, ra_hi[sl][i]);
        } else {
          split8(ra_lo[sl][i], ra_hi[sl][i], hi, lo);
        }
        *reinterpret_cast<uint4*>(&Ah[buf][r][ko]) = hi;
        *reinterpret_cast<uint4*>(&Al[buf][r][ko]) = lo;
      }
    }
#pragma unroll
    for (int i = 0; i < B_IT; i++) {
      const int idx = tid + i * NT, r = idx / (BK / 8), ko = (idx % (BK / 8)) * 8;
      if (r < BN) {
        *reinterpret_cast<uint4*>(&Bh[buf][r][ko]) = rbh[sl][i];
        *reinterpret_cast<uint4*>(&Bl[buf][r][ko]) = rbl[sl][i];
      }
    }
  };

  f32x16 acc[RM][RN], acl[RM][RN];                  // hi.hi, and hi.lo + lo.hi (scaled by 2^11)
#pragma unroll
  for (int i = 0; i < RM; i++)
#pragma unroll
    for (int j = 0; j < RN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) { acc[i][j][r] = 0.f; acl[i][j][r] = 0.f; }

  const int nk = max(0, kend - kbeg) / BK;
  auto compute_tile = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < BK / 16; ks++) {
      f16x8 ah[RM], al[RM], bh[RN], bl[RN];
#pragma unroll
      for (int i = 0; i < RM; i++) {
        ah[i] = *reinterpret_cast<const f16x8*>(&Ah[buf][wm * TM + i * 32 + li][ks * 16 + lh * 8]);
        al[i] = *reinterpret_cast<const f16x8*>(&Al[buf][wm * TM + i * 32 + li][ks * 16 + lh * 8]);
      }
#pragma unroll
      for (int j = 0; j < RN; j++) {
        bh[j] = *reinterpret_cast<const f16x8*>(&Bh[buf][wn * TN + j * 32 + li][ks * 16 + lh * 8]);
        bl[j] = *reinterpret_cast<const f16x8*>(&Bl[buf][wn * TN + j * 32 + li][ks * 16 + lh * 8]);
      }
      // (issue order: a term-major order -- all hi.hi, then all hi.lo, then all lo.hi, no two consecutive MFMAs on one
      // accumulator -- was measured in round 5 and changes nothing: large-v2 encoder 198.3 vs 198.3 ms)
#pragma unroll
      for (int i = 0; i < RM; i++)
#pragma unroll
        for (int j = 0; j < RN; j++) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          acl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acl[i][j], 0, 0, 0);
          acl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acl[i][j], 0, 0, 0);
        }
    }
  };
  using S0 = std::integral_constant<int, 0>;
  if constexpr (PF == 1) {
    if (nk > 0) {
      load_tile(S0{}, kbeg);
      store_tile(S0{}, 0);
    }
    __syncthreads();
    for (int t = 0; t < nk; t++) {
      const int buf = t & 1;
      if (t + 1 < nk) load_tile(S0{}, kbeg + (t + 1) * BK);
      compute_tile(buf);
      if (t + 1 < nk) store_tile(S0{}, buf ^ 1);
      __syncthreads();
    }
  } else {
    // tile j travels through register slot j % PF: tiles 0 .. PF - 1 are requested up front; a slot is refilled with tile
    // j + PF right after tile j has gone to LDS (loads return in order: storing tile j waits for tile j only)
    static_for<PF>([&](auto u) { if (decltype(u)::value < nk) load_tile(u, kbeg + decltype(u)::value * BK); });
    if (nk > 0) {
      store_tile(S0{}, 0);
      if (PF < nk) load_tile(S0{}, kbeg + PF * BK);
    }
    __syncthreads();
    for (int t0 = 0; t0 < nk; t0 += PF) {
      static_for<PF>([&](auto u) {
        constexpr int uu = decltype(u)::value;
        const int t = t0 + uu;
        if (t < nk) {                                // (block-uniform)
          const int buf = t & 1;
          compute_tile(buf);
          if (t + 1 < nk) {
            using SN = std::integral_constant<int, (uu + 1) % PF>;     // slot of tile t + 1 (t0 is a multiple of PF)
            store_tile(SN{}, buf ^ 1);
            if (t + 1 + PF < nk) load_tile(SN{}, kbeg + (t + 1 + PF) * BK);
          }
          __syncthreads();
        }
      });
    }
  }

  // epilogue: batched residual / positional reads from clamped addresses, then arithmetic, then predicated stores
  // (same structure as gemm.hip)
  bool bad = false;
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
        float v = (acc[i][j][r] + acl[i][j][r] * LO_UNSCALE) + bias;
        bad |= !(fabsf(v) < 3.0e38f);               // inf / NaN: an fp16 piece overflowed (|operand| >= 65504)
        if (g.act == ACT_GELU) v = gelu_erf(v);
        if (g.col_scale_period > 0) v *= cs;
        if (g.residual) v = res[r] + v;
        if (g.aux) v = v + ax[r];
        if (col_ok && row < M) {
          if (g.Ch) {                                // the consumer is another split-precision GEMM: pieces, not f32
            u16 ph, pl;
            split1(v, ph, pl);
            g.Ch[(int64_t)row * g.ldc + col] = ph; g.Cl[(int64_t)row * g.ldc + col] = pl;
          } else {
            Cout[cbase + (int64_t)row * g.ldc] = v;
          }
        }
      }
    }
  // range guard: the host re-runs the pass on the exact-f32 kernel and stops using this one (engine.cpp: split_guarded)
  if (bad && g.range_flag) __hip_atomic_fetch_or(g.range_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int BM, int BN, int WGM, int WGN, int PF>
void launch_cfg(hipStream_t st, const GemmArgs& a, const u16* Wh, const u16* Wl, int ldwt) {
  dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, a.ksplit > 1 ? a.ksplit : 1);
  if (a.Ah) WB_KLAUNCH((gemm_f16x3_kernel<BM, BN, WGM, WGN, PF, true>), grid, dim3(NT), 0, st, a, Wh, Wl, ldwt);
  else WB_KLAUNCH((gemm_f16x3_kernel<BM, BN, WGM, WGN, PF, false>), grid, dim3(NT), 0, st, a, Wh, Wl, ldwt);
}

// W [K][N] f32 -> hi, lo [N][K] fp16 (K-contiguous), through a 32 x 32 LDS tile
__global__ __launch_bounds__(256) void split_weight_f16_kernel(const float* __restrict__ W, int K, int N, u16* __restrict__ hi,
                                                               u16* __restrict__ lo) {
  __shared__ float t[32][33];
  const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int r = ty; r < 32; r += 8) t[r][tx] = (k0 + r < K && n0 + tx < N) ? W[(int64_t)(k0 + r) * N + n0 + tx] : 0.f;
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    if (n0 + r < N && k0 + tx < K) {
      u16 h, l;
      split1(t[tx][r], h, l);
      hi[(int64_t)(n0 + r) * K + k0 + tx] = h;
      lo[(int64_t)(n0 + r) * K + k0 + tx] = l;
    }
  }
}

}  // namespace

void launch_split_weight_f16(hipStream_t st, const float* W, int K, int N, uint16_t* hi, uint16_t* lo) {
  hipLaunchKernelGGL(split_weight_f16_kernel, dim3((N + 31) / 32, (K + 31) / 32), dim3(256), 0, st, W, K, N, hi, lo);
}

// a.B / a.ldb are ignored: the weight comes pre-split as Wh, Wl [N][ldwt] fp16 (K-contiguous).
int launch_gemm_f16x3(hipStream_t st, const GemmArgs& a, const uint16_t* Wh, const uint16_t* Wl, int ldwt) {
  if (a.M <= 0 || a.N <= 0) return 0;
  if (a.K % BK != 0 || ldwt % 8 != 0 || a.conv1_tstride > 0) return -1;
  if (a.Ah && (!a.Al || a.a_desc || a.lda % 8 != 0)) return -1;                 // pieces: plain 16-byte-aligned rows only
  if (a.Ch && (!a.Cl || a.ksplit > 1 || a.c_block_cols > 0)) return -1;
  if (a.ksplit > 1 && (a.bias || a.residual || a.aux || a.act != ACT_NONE || a.col_scale_period > 0)) return -1;
  auto blocks = [&](int bm, int bn) { return (int64_t)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); };
  static const int force_tile = []() { const char* e = getenv("WHISPER_HIP_SPLIT_TILE"); return e ? atoi(e) : 0; }();   // developer A/B
  // (measured in round 5 and not kept: 128 x 64 tiles with three / four k-tiles in flight -- half the accumulators buy the
  // registers for a deeper queue -- large-v2 encoder 226 / 225 ms against 195.6 for 128 x 128 with one: the large shapes are not
  // bound by bytes in flight; the extra operand traffic and A-split work per flop cost more: profiles/r05_i_k12_tiles.txt)
  if (a.M <= 32) launch_cfg<32, 128, 1, 4, WB_F16X3_PF_SMALL>(st, a, Wh, Wl, ldwt);
  else if (force_tile == 64 || (force_tile != 128 && (a.ksplit > 1 || blocks(128, 128) < 384))) launch_cfg<64, 64, 2, 2, WB_F16X3_PF_SMALL>(st, a, Wh, Wl, ldwt);
  else launch_cfg<128, 128, 2, 2, 1>(st, a, Wh, Wl, ldwt);
  return 0;
}

}  // namespace wb
