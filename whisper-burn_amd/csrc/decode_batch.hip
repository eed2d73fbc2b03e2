// Batch-mode decode (9 - 64 live rows): the skinny split-K weight-stream GEMM.
//
// A decode step with n rows multiplies [n x K] activations by every decoder weight once: the weights (20 MB for
// large-v2's QKV) are the traffic, the rows are few.  The encoder's 32 x 128 LDS-staged MFMA GEMM (gemm.hip) ran these
// at 13.9 us per launch / 0.18 of HBM peak (profiles/r03_a_bench_large_v2_450s.json): a block of it walks its K slice
// tile by tile behind barriers, so little of the matrix is in flight at any time.
//
// Here  (Linear: y = x W + b, W stored [d_in, d_out]; mod.rs:377-379, :429-435, :483-489)
//   * a wave owns a 64-column strip of W over its own K range and requests ALL of its rows before the first use
//     (one float4 per lane = 4 K-rows x 64 columns per load instruction; <= 20 instructions in flight per wave, 12 with
//     three or four row tiles): the
//     whole matrix is in flight one round trip after the launch;
//   * arithmetic stays exact f32 on the matrix cores: v_mfma_f32_16x16x4_f32, rows in tiles of 16 (e.g. 15 windows of
//     large-v2 fill one tile; 32 x 32 tiles would idle half the array).  Component c of the lane's float4 is the B
//     operand of accumulator c, i.e. accumulator c holds columns 4 j + c: no LDS staging for W, and a lane ends up
//     with a float4 of consecutive output columns;
//   * the 8 waves of a block share the strip and split the block's K slice; their partial tiles meet in LDS (fixed
//     wave order) and one split-K plane per block goes out -- the same planes the tiled GEMM wrote, folded by the same
//     consumers in the same order;
// 11.4 us per launch (0.20 of HBM peak) against 13.8 us on large-v2, 450 s (38 rows): 424x -> 450x on one box
// (profiles/r03_t_*); small, 10 min: 2330x -> 2525x (profiles/r03_p_*).
//
// Rejected, both measured on large-v2 450 s (the fold launches stay: a kernel boundary is ~7.6 us here, and every way
// tried of moving a fold across it cost more):
//   * "last arriver" epilogues -- the block that takes the last ticket of its strip sums the strip's planes (+ GELU), the
//     block that takes the last ticket of the launch normalises the rows (Guideline 16 R1 hand-offs: write-through stores,
//     drained waves, one agent-scope atomic).  Removes 4 of 12 launches per layer and still loses: every level costs three
//     dependent fabric round trips under full load (drain, ticket, plane re-read) -- 22 us for GEMM + GELU fold, 40 us
//     for GEMM + resolve + LayerNorm, against 11 + 4.3 and 11 + 6.7 (+ a boundary each): 374x (profiles/r03_h_*, r03_i_*);
//   * a GELU prologue in the MLP's second product (A = GELU(b1 + the first product's planes) formed while the block
//     stages its K slice): the planes are four times the activations and arrive in front of the weights -- that launch
//     became 16 us slower to save 4.3 us + a boundary: 441x (profiles/r03_j_*).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "decode.h"
#include "handoff.h"
#include "kernels.h"
#include "wave_ops.h"

namespace wb {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SK_NT = 512;          // 8 waves: one 64-column strip, the block's K slice in 8 parts
constexpr int SK_MAX_LD = 20;       // float4 loads a wave keeps in flight (80 K-rows); 16 with four row tiles
// loads in flight per wave by row tiles: the activations of a block's K slice (16 MT rows x 32 LDW K-rows) share the LDS with
// the eight waves' partial tiles; three row tiles with 20 loads are 120 KB (one block per CU either way: > 128 registers)
__host__ __device__ constexpr int sk_ldw(int mt) { return mt <= 3 ? SK_MAX_LD : 16; }

// PAIR (opt-in, WHISPER_HIP_SK_PAIR=1; to be measured): the partial tiles meet pairwise -- waves 4-7 park theirs, waves 0-3 add
// their own and park the sums, the plane items sum those four: half the LDS (16 KB per row tile instead of 32), so that a
// block of THREE row tiles is 74 KB and two of them share a CU.  Today's 96 KB admit one, and the 320 blocks of large-v2's
// MLP products (the smallest K-split 12 loads in flight allow) take two rounds of the chip: 19.5 us against 8.8 us for a bare
// read of the matrix (profiles/r03_m_layer_cycle_large_v2.txt, r03_v_stream_probe.txt).  The summation order becomes
// (w0 + w4) + (w1 + w5) + (w2 + w6) + (w3 + w7): fixed, but other bits than the sequential order's -- which is why it waits
// for a whole `-m gpu` run.
template <int MT, bool PAIR>
__global__ __launch_bounds__(SK_NT, MT <= 2 ? 2 : 1) void dec_skinny_gemm_kernel(SkinnyArgs a) {
  // one region, two lives: the activations of the block's K slice, k-major in row tiles of 16 (As[mt][k][16]: the A
  // operand of lane l for K-rows k0 .. k0 + 3 is word 16 k0 + l -- conflict-free), then the 8 waves' partial tiles
  // (K-rows per block <= 32 x 20 = 640: 10240 words per row tile -- 32 x 16 = 512 rows with four row tiles;
  // the partial tiles need 8 x 16 x 64 = 8192 words per row tile)
  // (round 3 kept 12 loads in flight with three / four row tiles: large-v2's MLP products then needed 320 blocks = TWO rounds
  // of the chip at one block per CU, 19.5 us against 11.2 us for the QKV product's 240 blocks -- profiles/r03_m_*; with 20 / 16
  // loads the same products are 160 - 200 blocks with every byte in flight from the start)
  constexpr int LDW = sk_ldw(MT);
  __shared__ __attribute__((aligned(16))) float smem[MT * 512 * LDW > MT * (PAIR ? 4096 : 8192) ? MT * 512 * LDW : MT * (PAIR ? 4096 : 8192)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int strip = blockIdx.x, z = blockIdx.y;
  // K-slice z: the 32-row tiles [z T / ks, (z + 1) T / ks) of the T = K / 32 tiles -- slices may differ by one tile, so that
  // the launch can have close to one block per compute unit whatever K / 32 factors into (round 6: large-v2's MLP products
  // ran 160 blocks on 256 CUs because 1280 / 32 = 40 tiles only split evenly by 2 or 4)
  const int nt_all = a.K >> 5;
  const int t0 = (int)((int64_t)z * nt_all / a.ksplit), t1 = (int)((int64_t)(z + 1) * nt_all / a.ksplit);
  const int kchunk = (t1 - t0) << 5;                 // host: every slice <= 32 LDW rows
  const int kw = kchunk >> 3;                        // K-rows per wave (multiple of 4)
  const int kb = t0 << 5;
  const int n0 = strip * 64;
  const int krow = lane >> 4, cq = lane & 15;

  // ---- requested first (loads return in order): the activations of the K slice (rows past M are zeros) ...
  constexpr int AQ = (MT * 16 * (32 * LDW / 4) + SK_NT - 1) / SK_NT;    // float4 quads per thread, at most
  const int nq = kchunk >> 2;                        // float4 quads per row
  float4 av4[AQ];
#pragma unroll
  for (int i = 0; i < AQ; i++) {
    const int idx = tid + i * SK_NT, r = idx % (MT * 16), kq = idx / (MT * 16);
    av4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kq < nq && r < a.M) av4[i] = *reinterpret_cast<const float4*>(a.A + (int64_t)r * a.lda + kb + 4 * kq);
  }
  // ---- ... then the wave's weights: every row requested now
  const int nld = kw >> 2;
  float4 bw[LDW];
  {
    const float* bp = a.B + (int64_t)(kb + wave * kw + krow) * a.ldb + n0 + 4 * cq;
#pragma unroll
    for (int t = 0; t < LDW; t++)
      if (t < nld) bw[t] = *reinterpret_cast<const float4*>(bp + (int64_t)(4 * t) * a.ldb);
  }
#pragma unroll
  for (int i = 0; i < AQ; i++) {
    const int idx = tid + i * SK_NT, r = idx % (MT * 16), kq = idx / (MT * 16);
    if (kq < nq) {
      float* dst = smem + ((r >> 4) * kchunk + 4 * kq) * 16 + (r & 15);
      dst[0] = av4[i].x; dst[16] = av4[i].y; dst[32] = av4[i].z; dst[48] = av4[i].w;
    }
  }
  __syncthreads();
  f32x4 acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int c = 0; c < 4; c++) acc[mt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const float* ap = smem + (wave * kw) * 16 + lane;
#pragma unroll
    for (int t = 0; t < LDW; t++) {
      if (t < nld) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
          const float av = ap[(mt * kchunk + 4 * t) * 16];
          acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw[t].x, acc[mt][0], 0, 0, 0);
          acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw[t].y, acc[mt][1], 0, 0, 0);
          acc[mt][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw[t].z, acc[mt][2], 0, 0, 0);
          acc[mt][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw[t].w, acc[mt][3], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();                                   // the activations are consumed: the region takes the partial tiles
  // red[wave][mt][i][j] as float4 (columns 4 j .. 4 j + 3): register v of lane l is row 4 (l / 16) + v, column j = l % 16
  constexpr int NSLOT = PAIR ? 4 : 8;
  auto park = [&](int slot) {
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int i = 4 * krow + v;
        *reinterpret_cast<float4*>(smem + (((slot * MT + mt) * 16 + i) * 16 + cq) * 4) =
            make_float4(acc[mt][0][v], acc[mt][1][v], acc[mt][2][v], acc[mt][3][v]);
      }
  };
  if constexpr (PAIR) {
    if (wave >= 4) park(wave - 4);
    __syncthreads();
    if (wave < 4) {                                    // (a wave reads and rewrites its OWN slot: no barrier in between)
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int v = 0; v < 4; v++) {
          const int i = 4 * krow + v;
          const float4 t = *reinterpret_cast<const float4*>(smem + (((wave * MT + mt) * 16 + i) * 16 + cq) * 4);
          acc[mt][0][v] += t.x; acc[mt][1][v] += t.y; acc[mt][2][v] += t.z; acc[mt][3][v] += t.w;
        }
      park(wave);
    }
  } else {
    park(wave);
  }
  __syncthreads();
  // ---- the block's plane: (row, column quad) items over the block's 512 threads -- 16 MT rows x 16 quads, i.e. up to two
  // items per thread with three or four row tiles -- each sums the waves in order
#pragma unroll
  for (int e0 = 0; e0 < MT * 256; e0 += SK_NT) {
    const int e = e0 + tid, row = e >> 4, j4 = e & 15;
    if (e < MT * 256 && row < a.M) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int w = 0; w < NSLOT; w++) {
        const float4 t = *reinterpret_cast<const float4*>(smem + ((w * MT * 16 + row) * 16 + j4) * 4);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
      *reinterpret_cast<float4*>(a.P + (int64_t)z * a.plane + (int64_t)row * a.N + n0 + 4 * j4) = s;
    }
  }
}


// ---- split-precision variant (round 5): the same launch geometry, planes and consumers, arithmetic on the 16-bit matrix path
//
// At 33 - 64 rows the exact-f32 MFMA is the launch's longest phase, not the weight stream: a wave issues LDW x MT x 4 =
// 240 v_mfma_f32_16x16x4_f32 (32 clk each) at three row tiles, two waves share a SIMD: 15 360 clk = 6.4 us of matrix-pipe time
// per block inside a launch of 10.5 us on average (large-v2, 38 rows: profiles/r04_e_bench_default.json).  Here the weight
// arrives as fp16 hi / lo (x 2^11) pieces -- the same 4 bytes per element, so the stream is what it was -- and the
// activations are split on their way into LDS (A = A_hi + 2^-11 A_lo, as gemm_f16x3.hip):
//   A W  ~=  A_hi W_hi + 2^-11 (A_hi W_lo + A_lo W_hi)        three v_mfma_f32_16x16x32_f16 (16 clk each) per 32-deep tile
// i.e. 108 MFMAs of half the duration per wave: 1.4 us.  f32 accumulation, 22-bit operands: f32-grade (and each output's K
// chain is 8x shorter than the f32 kernel's, which is what set the distance to the exact result in round 4, LABLOG R4.1).
//
// Weight layout (LinearW::th / tl, made at load by split_weight_f16_tiles): [n / 16][k / 32][kg = (k % 32) / 8][j = n % 16][8]
// -- the B operand of lane l = 16 kg + j for one (16-column, 32-deep) tile is the 16 bytes at l x 16 of a 1 KB block: every
// load instruction of a wave is one fully coalesced kilobyte.  Activations in LDS the same way ([row tile][k / 32][kg][i][8]),
// so the A operand is one conflict-free ds_read_b128.  A wave owns the strip's four 16-column tiles over its share of the
// block's 32-deep K tiles (the block's <= 20 tiles dealt 3 / 2 to the eight waves), all of them requested before the first use.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
constexpr int SKH_MAXCH = 3;        // 32-deep K tiles per wave (kchunk <= 640 = 20 tiles over 8 waves)
constexpr float SKH_LO_SCALE = 2048.f, SKH_LO_UNSCALE = 1.f / 2048.f;

__device__ __forceinline__ void skh_split4(const float4 v, uint2& hi, uint2& lo) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  u16 h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const _Float16 hh = (_Float16)x[i];
    h[i] = __builtin_bit_cast(u16, hh);
    l[i] = __builtin_bit_cast(u16, (_Float16)((x[i] - (float)hh) * SKH_LO_SCALE));
  }
  hi = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
  lo = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
}

template <int MT>
__global__ __launch_bounds__(SK_NT, MT <= 2 ? 2 : 1) void dec_skinny_f16x3_kernel(SkinnyArgs a) {
  // one region, two lives: the activations of the block's K slice as fp16 pieces (hi then lo: MT x kchunk x 16 halves each),
  // then the 8 waves' partial tiles (8 x MT x 16 x 64 floats)
  constexpr int LDW = sk_ldw(MT);
  constexpr int WORDS = MT * 512 * LDW > MT * 8192 ? MT * 512 * LDW : MT * 8192;
  __shared__ __attribute__((aligned(16))) float smem[WORDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int strip = blockIdx.x, z = blockIdx.y;
  const int nt_all = a.K >> 5;                       // (uneven K slices: see dec_skinny_gemm_kernel)
  const int t0 = (int)((int64_t)z * nt_all / a.ksplit), t1 = (int)((int64_t)(z + 1) * nt_all / a.ksplit);
  const int kchunk = (t1 - t0) << 5;                 // host: every slice <= 32 LDW rows
  const int nch = t1 - t0;                           // 32-deep tiles of the block
  const int kb = t0 << 5;
  const int n0 = strip * 64;
  const int kg = lane >> 4, j = lane & 15;
  u16* As_hi = reinterpret_cast<u16*>(smem);
  u16* As_lo = As_hi + MT * kchunk * 16;

  // ---- requested first (loads return in order): the activations of the K slice (rows past M are zeros) ...
  constexpr int AQ = (MT * 16 * (32 * LDW / 4) + SK_NT - 1) / SK_NT;    // float4 quads per thread, at most
  const int nq = kchunk >> 2;
  float4 av4[AQ];
#pragma unroll
  for (int i = 0; i < AQ; i++) {
    const int idx = tid + i * SK_NT, r = idx % (MT * 16), kq = idx / (MT * 16);
    av4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kq < nq && r < a.M) av4[i] = *reinterpret_cast<const float4*>(a.A + (int64_t)r * a.lda + kb + 4 * kq);
  }
  // ---- ... then the wave's weight tiles: every one requested now (1 KB per instruction, linear in the lane)
  const int c0 = wave * nch / 8, c1 = (wave + 1) * nch / 8;
  f16x8 bh[SKH_MAXCH][4], bl[SKH_MAXCH][4];
  {
    const int64_t ktiles = a.K >> 5;
#pragma unroll
    for (int t = 0; t < SKH_MAXCH; t++)
      if (c0 + t < c1) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const int64_t off = (((int64_t)(n0 / 16 + c) * ktiles + (kb >> 5) + c0 + t) * 64 + lane) * 8;
          bh[t][c] = *reinterpret_cast<const f16x8*>(a.Bh + off);
          bl[t][c] = *reinterpret_cast<const f16x8*>(a.Bl + off);
        }
      }
  }
#pragma unroll
  for (int i = 0; i < AQ; i++) {
    const int idx = tid + i * SK_NT, r = idx % (MT * 16), kq = idx / (MT * 16);
    if (kq < nq) {
      const int k = 4 * kq;                          // k within the slice: tile k / 32, octet (k % 32) / 8, half k % 8 in {0, 4}
      const int o = ((((r >> 4) * nch + (k >> 5)) * 4 + ((k & 31) >> 3)) * 16 + (r & 15)) * 8 + (k & 7);
      uint2 hi, lo;
      skh_split4(av4[i], hi, lo);
      *reinterpret_cast<uint2*>(As_hi + o) = hi;
      *reinterpret_cast<uint2*>(As_lo + o) = lo;
    }
  }
  __syncthreads();
  f32x4 acc[MT][4], acl[MT][4];                      // hi.hi, and hi.lo + lo.hi (scaled by 2^11)
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int c = 0; c < 4; c++) { acc[mt][c] = f32x4{0.f, 0.f, 0.f, 0.f}; acl[mt][c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int t = 0; t < SKH_MAXCH; t++)
    if (c0 + t < c1) {
#pragma unroll
      for (int mt = 0; mt < MT; mt++) {
        const int o = ((mt * nch + c0 + t) * 64 + lane) * 8;
        const f16x8 ah = *reinterpret_cast<const f16x8*>(As_hi + o);
        const f16x8 al = *reinterpret_cast<const f16x8*>(As_lo + o);
#pragma unroll
        for (int c = 0; c < 4; c++) {
          acc[mt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[t][c], acc[mt][c], 0, 0, 0);
          acl[mt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[t][c], acl[mt][c], 0, 0, 0);
          acl[mt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[t][c], acl[mt][c], 0, 0, 0);
        }
      }
    }
  __syncthreads();                                   // the activations are consumed: the region takes the partial tiles
  // red[wave][mt][i][col]: register v of lane l is row i = 4 (l / 16) + v, column 16 c + l % 16 of the strip
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int v = 0; v < 4; v++)
        smem[((wave * MT + mt) * 16 + 4 * kg + v) * 64 + 16 * c + j] = acc[mt][c][v] + acl[mt][c][v] * SKH_LO_UNSCALE;
  __syncthreads();
  // ---- the block's plane: (row, column quad) items over the block's 512 threads, each sums the waves in order
  const int n_live = a.st ? a.st[ST_N] : a.M;        // (rows past the live count are launched with whatever their buffers hold)
  bool bad = false;
#pragma unroll
  for (int e0 = 0; e0 < MT * 256; e0 += SK_NT) {
    const int e = e0 + tid, row = e >> 4, j4 = e & 15;
    if (e < MT * 256 && row < a.M) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int w = 0; w < 8; w++) {
        const float4 t = *reinterpret_cast<const float4*>(smem + ((w * MT * 16 + row) * 16 + j4) * 4);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
      if (row < n_live)
        bad |= !(fabsf(s.x) < 3.0e38f) || !(fabsf(s.y) < 3.0e38f) || !(fabsf(s.z) < 3.0e38f) || !(fabsf(s.w) < 3.0e38f);
      *reinterpret_cast<float4*>(a.P + (int64_t)z * a.plane + (int64_t)row * a.N + n0 + 4 * j4) = s;
    }
  }
  // range guard (an activation left fp16's range: inf / NaN): the host fails the decode call and switches the model to the
  // exact-f32 kernel above (session.cpp: dec_split_check)
  if (bad && a.range_flag) __hip_atomic_fetch_or(a.range_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// W [K][N] f32 -> hi, lo tiles [N / 16][K / 32][(k % 32) / 8][n % 16][k % 8]: one block per (16 columns, 32 K-rows) tile
__global__ __launch_bounds__(256) void split_weight_f16_tiles_kernel(const float* __restrict__ W, int K, int N,
                                                                     u16* __restrict__ hi, u16* __restrict__ lo) {
  const int tn = blockIdx.x, tk = blockIdx.y;
  for (int e = threadIdx.x; e < 512; e += 256) {
    const int kk = e >> 4, jj = e & 15;              // reads: 16 consecutive columns of one K-row
    const float x = W[(int64_t)(tk * 32 + kk) * N + tn * 16 + jj];
    const _Float16 h = (_Float16)x;
    const int64_t o = ((int64_t)tn * (K >> 5) + tk) * 512 + (((kk >> 3) * 16 + jj) * 8 + (kk & 7));
    hi[o] = __builtin_bit_cast(u16, h);
    lo[o] = __builtin_bit_cast(u16, (_Float16)((x - (float)h) * SKH_LO_SCALE));
  }
}

}  // namespace

// K-splits of the skinny GEMM: slices a block's 8 waves can share in multiples of 4 rows (K % (32 ks) == 0), at most
// SK_MAX_LD float4 rows in flight per wave, and as many blocks as fit ONE round of the chip: with 320 blocks on 256 CUs the
// MLP products took 19.5 us against 11.2 us for the QKV product's 240 blocks of the same size
// (profiles/r03_l_layer_cycle_large_v2.txt).  0: the shape is not served (the caller keeps the tiled GEMM)
int skinny_ksplit(int K, int N, int max_ks, int max_rows) {
  if (N % 64 != 0 || K % 32 != 0) return 0;
  static const int max_blocks = []() { const char* e = getenv("WHISPER_HIP_SK_MAX_BLOCKS"); return e ? atoi(e) : 256; }();
  const int strips = N / 64;
  int best = 0;
  const int nt = K / 32;
  for (int ks = 1; ks <= std::min(std::min(KS_MAX, max_ks), nt); ks++) {
    if ((nt + ks - 1) / ks > sk_ldw((max_rows + 15) / 16)) continue;      // the largest slice (slices differ by <= one 32-row tile)
    if (best > 0 && strips * ks > max_blocks) break;
    best = ks;
  }
  return best;
}

bool skinny_supported(int M, int K, int N) { return M >= 1 && M <= 64 && skinny_ksplit(K, N, KS_MAX, M) > 0; }

int launch_dec_skinny_gemm(hipStream_t st, const SkinnyArgs& a) {
  if (a.ksplit < 1 || a.ksplit > KS_MAX || a.K % 32 != 0 || a.ksplit > a.K / 32 || a.N % 64 != 0 || a.M < 1 || a.M > 64) return -1;
  if ((a.K / 32 + a.ksplit - 1) / a.ksplit > sk_ldw((a.M + 15) / 16)) return -1;
  const int MT = (a.M + 15) / 16;
  const dim3 grid(a.N / 64, a.ksplit), block(SK_NT);
  if (a.Bh && a.Bl) {                                // split-precision variant: same geometry, same planes
    switch (MT) {
      case 1: WB_KLAUNCH((dec_skinny_f16x3_kernel<1>), grid, block, 0, st, a); break;
      case 2: WB_KLAUNCH((dec_skinny_f16x3_kernel<2>), grid, block, 0, st, a); break;
      case 3: WB_KLAUNCH((dec_skinny_f16x3_kernel<3>), grid, block, 0, st, a); break;
      default: WB_KLAUNCH((dec_skinny_f16x3_kernel<4>), grid, block, 0, st, a); break;
    }
    return 0;
  }
  static const bool pair = []() { const char* e = getenv("WHISPER_HIP_SK_PAIR"); return e && e[0] == '1'; }();
#define WB_SK(MT_)                                                                              \
  do {                                                                                          \
    if (pair) WB_KLAUNCH((dec_skinny_gemm_kernel<MT_, true>), grid, block, 0, st, a);            \
    else WB_KLAUNCH((dec_skinny_gemm_kernel<MT_, false>), grid, block, 0, st, a);                \
  } while (0)
  switch (MT) {
    case 1: WB_SK(1); break;
    case 2: WB_SK(2); break;
    case 3: WB_SK(3); break;
    default: WB_SK(4); break;
  }
#undef WB_SK
  return 0;
}

void launch_split_weight_f16_tiles(hipStream_t st, const float* W, int K, int N, uint16_t* hi, uint16_t* lo) {
  hipLaunchKernelGGL(split_weight_f16_tiles_kernel, dim3(N / 16, K / 32), dim3(256), 0, st, W, K, N, hi, lo);
}

}  // namespace wb
