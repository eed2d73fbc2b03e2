// K1: fused log-mel frontend for gfx950.
//
// Replaces /root/reference/src/audio.rs:34-56 (prep_audio) + :284-367 (stfft: dense DFT by
// two [201x400]x[400xT] f32 matmuls, ~60 tiny launches and 3 blocking D2H reads per window)
// with one launch per batch of windows plus a clamp fix-up launch that normally touches no tile:
//   stage 0  interior blocks copy their 32 frame rows as 16-byte pieces (overlaps served by L1); blocks at a
//            window edge read their contiguous PCM span (5360 samples) once with reflect indexing
//            (audio.rs:297-306; no materialised padded copy) and scatter each sample to its <= 3 frame rows
//   stage 1  Hann (audio.rs:272-278, per-lane registers) + 20-point DFTs over n1   } 400-point DFT as an
//   stage 2  twiddles W400^{n2 k1}, transpose through LDS, 20-point DFTs over n2   } LDS-staged 20x20
//            Cooley-Tukey FFT, two real frames packed into one complex transform
//   stage 3  unpack the two spectra, |X|^2 for bins 0..200
//   stage 4  sparse Slaney filterbank (<= 16 taps per row, audio.rs:67-143; lane = frame, half-wave = 8
//            mel rows; taps, starts and lengths sit in LDS behind the power spectra so the tap loop is
//            unpredicated 4-tap chunks of broadcast LDS reads), relu(x-1e-10)+1e-10, ln(x)/ln10
//            (helper.rs:8-10, :24-27), block maximum (no atomics)
//   stage 5  16-byte stores of the [80][32] tile
//            of (x + 4) / 4 (audio.rs:53); zero padding frames (transcribe.rs:171-177)
// fix-up:  window max over the tile maxima (audio.rs:50); tiles whose minimum is below max - 8 are clamped (audio.rs:52).
// Bound: HBM (640 B PCM in + 320 B mel out per frame); ~11 kFLOP per frame of f32 VALU.
//
// Geometry: block = 320 threads = 16 pairs x 20 lanes; a pair transforms frames (2p, 2p+1) of the
// block's 32 consecutive frames; lane q of a pair runs one 20-point complex DFT per stage in
// registers.  Every stage aliases the same 16 x 852-float pair-private LDS regions (54.5 KB: three
// blocks per CU); row strides are chosen from the bank maps (2 x 426 = 852 = 20 mod 32 -> bank = tid).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include "kernels.h"
#include "wave_ops.h"

namespace wb {

namespace {

constexpr int FPB = 32;          // frames per block
constexpr int PAIRS = FPB / 2;   // 16
constexpr int MEL_THREADS = PAIRS * 20;   // 320
constexpr int FROW = 426;        // LDS floats per frame row: 2*FROW = 852 >= 840 (U) and 852 % 32 == 20
constexpr int UROW = 21;         // padded row (float2) of the 20x20 intermediate
constexpr int PB_OFF = 209;      // frame b's power spectrum: 209 = 17 mod 64, so stage 4's lane = frame reads hit 32 distinct banks
constexpr int OT_OFF = 416;      // a pair's 2 x 80 outputs live behind its two power spectra
constexpr int TAP_OFF = 576;     // ... and behind them the taps of mel rows 5p .. 5p+4 (5 x 16 floats) for stage 4
constexpr int BMAX_OFF = 800;    // the block-maximum scratch (pair 0's region)

struct cpx { float re, im; };

// (developer A/B, WB_MEL_SETTLE: idle cycles between the instruction that produced a value and the LDS store that reads it)
#if defined(WB_MEL_SETTLE) && !defined(HIPEMU)
#define WB_SETTLE2(a, b) asm volatile("s_nop 3" : "+v"(a), "+v"(b))
#else
#define WB_SETTLE2(a, b) do { } while (0)
#endif

__device__ __forceinline__ cpx cmul(cpx a, float wr, float wi) {
  return {a.re * wr - a.im * wi, a.re * wi + a.im * wr};
}

#ifndef WB_MEL_MIN_BLOCKS
#define WB_MEL_MIN_BLOCKS 4
#endif

// ---- 20-point forward DFT (e^{-i...}) in registers: 5 radix-4 butterflies, twiddles W20^{bc}, 4 radix-5 butterflies
// (n = 5a + b, k = c + 4e).  Every complex value is ONE 64-bit register pair and every butterfly ONE packed instruction: the
// half-swaps and sign flips of "multiply by -i" / complex multiplication by a constant ride in the VOP3P op_sel / neg
// modifiers instead of register moves.  (The plain-C form of rounds 1-4 left the packing to the compiler's SLP pass, which
// built the swapped pairs with moves: 892 VALU instructions for the two transforms of a thread, 197 of them v_mov, against
// 572 / 54 here.  Measured in round 5 on one box, alternating: 1.14 -> 1.21 G frames/s, with the plain v_fmac filterbank
// sums below 1.30 -- profiles/r05_a_variants.txt.)
typedef float f2 __attribute__((ext_vector_type(2)));
#if defined(HIPEMU) || defined(WB_MEL_NO_ASM)
#if defined(HIPEMU)
#define WB_PK static inline
#else
#define WB_PK __device__ __forceinline__   // (developer A/B: the arithmetic without the hand-written instructions)
#endif
// functional model: the helpers by their arithmetic (mul then fma, as the two-instruction forms below round)
#if defined(WB_MEL_NO_PK)   // (developer A/B, with -fno-slp-vectorize: no packed-FP32 instruction at all)
WB_PK f2 pk_add(f2 a, f2 b) { return f2{a.x + b.x, a.y + b.y}; }
WB_PK f2 pk_sub(f2 a, f2 b) { return f2{a.x - b.x, a.y - b.y}; }
#else
WB_PK f2 pk_add(f2 a, f2 b) { return a + b; }
WB_PK f2 pk_sub(f2 a, f2 b) { return a - b; }
#endif
WB_PK f2 pk_add_mi(f2 a, f2 b) { return f2{a.x + b.y, a.y - b.x}; }          // a - i b
WB_PK f2 pk_add_pi(f2 a, f2 b) { return f2{a.x - b.y, a.y + b.x}; }          // a + i b
WB_PK f2 pk_cmul(f2 a, f2 w) {                                                 // a * (w.x + i w.y)
  f2 r = f2{a.x * w.x, a.y * w.x};
  return f2{fmaf(-a.y, w.y, r.x), fmaf(a.x, w.y, r.y)};
}
WB_PK f2 pk_fma_lo(f2 a, f2 c, f2 acc) { return f2{fmaf(a.x, c.x, acc.x), fmaf(a.y, c.x, acc.y)}; }   // acc + a * c.x
WB_PK f2 pk_fma_hi(f2 a, f2 c, f2 acc) { return f2{fmaf(a.x, c.y, acc.x), fmaf(a.y, c.y, acc.y)}; }   // acc + a * c.y
WB_PK f2 pk_mul_lo(f2 a, f2 c) { return f2{a.x * c.x, a.y * c.x}; }
WB_PK f2 pk_mul_hi(f2 a, f2 c) { return f2{a.x * c.y, a.y * c.y}; }
#else
#if defined(WB_MEL_EARLYCLOBBER)
#define WB_PK_OUT "=&v"
#else
#define WB_PK_OUT "=v"
#endif
__device__ __forceinline__ f2 pk_add(f2 a, f2 b) { return a + b; }
__device__ __forceinline__ f2 pk_sub(f2 a, f2 b) { return a - b; }
// a - i b = (a.x + b.y, a.y - b.x): LO = src0.lo + src1.hi, HI = src0.hi - src1.lo
__device__ __forceinline__ f2 pk_add_mi(f2 a, f2 b) {
  f2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : WB_PK_OUT(r) : "v"(a), "v"(b));
  return r;
}
// a + i b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ f2 pk_add_pi(f2 a, f2 b) {
  f2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : WB_PK_OUT(r) : "v"(a), "v"(b));
  return r;
}
// a * (w.x + i w.y) = (a.x w.x - a.y w.y, a.y w.x + a.x w.y): two instructions, the twiddle one 64-bit constant pair
__device__ __forceinline__ f2 pk_cmul(f2 a, f2 w) {
  f2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : WB_PK_OUT(r) : "v"(a), "v"(w));                 // (a.x w.x, a.y w.x)
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "+v"(r) : "v"(a), "v"(w));   // LO -= a.y w.y; HI += a.x w.y
  return r;
}
// acc + a * c.x / acc + a * c.y (a complex value times one REAL constant of the pair c, both halves)
__device__ __forceinline__ f2 pk_fma_lo(f2 a, f2 c, f2 acc) {
  f2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : WB_PK_OUT(r) : "v"(a), "v"(c), "v"(acc));
  return r;
}
__device__ __forceinline__ f2 pk_fma_hi(f2 a, f2 c, f2 acc) {
  f2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : WB_PK_OUT(r) : "v"(a), "v"(c), "v"(acc));
  return r;
}
__device__ __forceinline__ f2 pk_mul_lo(f2 a, f2 c) {
  f2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : WB_PK_OUT(r) : "v"(a), "v"(c));
  return r;
}
__device__ __forceinline__ f2 pk_mul_hi(f2 a, f2 c) {
  f2 r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : WB_PK_OUT(r) : "v"(a), "v"(c));
  return r;
}
#endif

__device__ __forceinline__ void dft20(cpx (&xc)[20]) {
  // W20^j = (cos(2 pi j / 20), -sin(2 pi j / 20)), j = 0..12 as (re, im) pairs
  const f2 W[13] = {{1.f, 0.f}, {0.95105651629515357f, -0.30901699437494742f}, {0.80901699437494742f, -0.58778525229247313f},
                    {0.58778525229247313f, -0.80901699437494742f}, {0.30901699437494742f, -0.95105651629515357f}, {0.f, -1.f},
                    {-0.30901699437494742f, -0.95105651629515357f}, {-0.58778525229247313f, -0.80901699437494742f},
                    {-0.80901699437494742f, -0.58778525229247313f}, {-0.95105651629515357f, -0.30901699437494742f}, {-1.f, 0.f},
                    {-0.95105651629515357f, 0.30901699437494742f}, {-0.80901699437494742f, 0.58778525229247313f}};
  const f2 C12 = {0.30901699437494742f, -0.80901699437494742f};     // (cos 2 pi / 5, cos 4 pi / 5)
  const f2 S12 = {0.95105651629515357f, 0.58778525229247313f};      // (sin 2 pi / 5, sin 4 pi / 5)
  f2 x[20], u[5][4];
#pragma unroll
  for (int i = 0; i < 20; i++) x[i] = f2{xc[i].re, xc[i].im};
#pragma unroll
  for (int b = 0; b < 5; b++) {
    const f2 s02 = pk_add(x[b], x[10 + b]), d02 = pk_sub(x[b], x[10 + b]);
    const f2 s13 = pk_add(x[5 + b], x[15 + b]), d13 = pk_sub(x[5 + b], x[15 + b]);
    const f2 t0 = pk_add(s02, s13), t2 = pk_sub(s02, s13);
    const f2 t1 = pk_add_mi(d02, d13), t3 = pk_add_pi(d02, d13);   // d02 - i d13, d02 + i d13
    u[b][0] = t0;
    if (b == 0) { u[b][1] = t1; u[b][2] = t2; u[b][3] = t3; }
    else { u[b][1] = pk_cmul(t1, W[b]); u[b][2] = pk_cmul(t2, W[2 * b]); u[b][3] = pk_cmul(t3, W[3 * b]); }
  }
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const f2 u0 = u[0][c];
    const f2 a1 = pk_add(u[1][c], u[4][c]), a2 = pk_add(u[2][c], u[3][c]);
    const f2 b1 = pk_sub(u[1][c], u[4][c]), b2 = pk_sub(u[2][c], u[3][c]);
    const f2 p1 = pk_fma_hi(a2, C12, pk_fma_lo(a1, C12, u0));      // u0 + C1 a1 + C2 a2
    const f2 p2 = pk_fma_lo(a2, C12, pk_fma_hi(a1, C12, u0));      // u0 + C2 a1 + C1 a2
    const f2 q1 = pk_fma_hi(b2, S12, pk_mul_lo(b1, S12));          // S1 b1 + S2 b2
    const f2 q2 = pk_sub(pk_mul_hi(b1, S12), pk_mul_lo(b2, S12));  // S2 b1 - S1 b2
    x[c] = pk_add(pk_add(u0, a1), a2);
    x[c + 4] = pk_add_mi(p1, q1);                                   // p1 - i q1
    x[c + 16] = pk_add_pi(p1, q1);                                  // p1 + i q1
    x[c + 8] = pk_add_mi(p2, q2);
    x[c + 12] = pk_add_pi(p2, q2);
  }
#pragma unroll
  for (int i = 0; i < 20; i++) xc[i] = cpx{x[i].x, x[i].y};
}

__global__ __launch_bounds__(MEL_THREADS, WB_MEL_MIN_BLOCKS) void mel_spectrogram_kernel(
    const float* __restrict__ pcm, const MelWindow* __restrict__ wins, const MelTables* __restrict__ tabs,
    float* __restrict__ out, int64_t win_stride, int row_stride, float* __restrict__ gmax, int bmax_stride, int pad,
    int pad_limit) {
  // 16 x 852 floats = 54 528 B: three blocks per CU (163 584 B of the 160 KiB); every stage aliases the same
  // pair-private regions (frame rows -> U -> Z -> power spectra + this pair's 2 x 80 outputs)
  __shared__ __attribute__((aligned(16))) float lds[PAIRS * 2 * FROW];

  const MelWindow w = wins[blockIdx.y];
  const int f0 = blockIdx.x * FPB;
  const int tid = threadIdx.x;
  const int zend = min(w.n_emit + pad, pad_limit);            // frames [n_emit, zend) := 0  (transcribe.rs:171-177)
  if (f0 >= w.n_frames) {
    // past the last frame: at most the zero padding frames fall into this tile.  No (max, min) pair is written: the
    // fix-up pass only reads the tiles that hold real frames (blk < ceil(n_frames / FPB)), and with a large padding the
    // grid may be wider than the per-window stride of `gmax` (a pair written here could land in the next window's slots).
    float* oz = out + (int64_t)blockIdx.y * win_stride;
    for (int e = tid; e < MEL_N_MELS * FPB; e += MEL_THREADS) {
      const int m = e / FPB, f = f0 + (e - m * FPB);
      if (f >= w.n_emit && f < zend) oz[(int64_t)m * row_stride + f] = 0.f;
    }
    return;
  }
  const int N = w.n_samples;
  const float* x = pcm + w.pcm_off;
  static_assert(MEL_N_MELS * MEL_MAX_TAPS == MEL_THREADS * 4, "one float4 of taps per thread");

  // ---- stage 0: the block's contiguous PCM span (31 hops + 400 = 5360 samples) is read ONCE, coalesced,
  // with reflect indexing at the window edges (audio.rs:297-306); every sample is scattered to the <= 3
  // frame rows that contain it.  The Hann window is applied in stage 1 from per-lane registers.
  constexpr int SPAN = (FPB - 1) * MEL_HOP + MEL_N_FFT;   // 5360
  constexpr int S0_ITERS = (SPAN + MEL_THREADS - 1) / MEL_THREADS;   // 17
  const int g0 = f0 * MEL_HOP - MEL_N_FFT / 2;
  if (g0 >= 0 && g0 + SPAN <= N) {
    // interior block (all but the first and the last one or two of a window): no reflection, so every frame
    // row is 100 consecutive 16-byte pieces of the PCM -- straight copies, ten per thread (the overlapping
    // parts of neighbouring rows come from L1; HBM still sees each sample once)
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // window starts are sample-, not 16 B-aligned
    constexpr int ROW_Q = MEL_N_FFT / 4, S0F_ITERS = FPB * ROW_Q / MEL_THREADS;   // 100 pieces per row, 10 per thread
    static_assert(FPB * ROW_Q % MEL_THREADS == 0, "frame rows must divide over the block");
    f4u v[S0F_ITERS];
#pragma unroll
    for (int i = 0; i < S0F_ITERS; i++) {
      const int e = tid + i * MEL_THREADS, fr = e / ROW_Q, c4 = (e - fr * ROW_Q) * 4;
      v[i] = *reinterpret_cast<const f4u*>(x + g0 + fr * MEL_HOP + c4);
    }
#pragma unroll
    for (int i = 0; i < S0F_ITERS; i++) {
      const int e = tid + i * MEL_THREADS, fr = e / ROW_Q, c4 = (e - fr * ROW_Q) * 4;
      float2* dst = reinterpret_cast<float2*>(&lds[fr * FROW + c4]);      // FROW is even: 8-byte aligned
      dst[0] = make_float2(v[i].x, v[i].y);
      dst[1] = make_float2(v[i].z, v[i].w);
    }
  } else {
    float xv[S0_ITERS];
#pragma unroll
    for (int i = 0; i < S0_ITERS; i++) {
      const int g = tid + i * MEL_THREADS;
      int j = g0 + g;
      if (j < 0) j = -j;
      if (j >= N) j = 2 * (N - 1) - j;
      j = max(0, min(j, N - 1));   // frames past the window's last frame are never emitted
      xv[i] = g < SPAN ? x[j] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < S0_ITERS; i++) {
      const int g = tid + i * MEL_THREADS;
      if (g < SPAN) {
        const int fhi = min(g / MEL_HOP, FPB - 1);          // last frame that starts at or before g
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const int fr = fhi - k, n = g - fr * MEL_HOP;
          if (fr >= 0 && n < MEL_N_FFT) lds[fr * FROW + n] = xv[i];
        }
      }
    }
  }
  __syncthreads();

  const int p = tid / 20, q = tid - p * 20;
  float* reg = lds + p * 2 * FROW;   // this pair's private region
  // this lane's constants (twiddles W400^{q k1}, filterbank rows q, q+20, q+40, q+60): one batch of loads
  // whose latency hides under the first DFT
  float2 twv[20];
#pragma unroll
  for (int k1 = 0; k1 < 20; k1++) twv[k1] = tabs->tw[q * 20 + k1];
  cpx z[20];
  // ---- stage 1: 20-point DFT over n1 of z[n1] = xa[20 n1 + q] + i xb[20 n1 + q] ----
#pragma unroll
  for (int n1 = 0; n1 < 20; n1++) {
    const float hw = tabs->hann[20 * n1 + q];   // periodic Hann (audio.rs:272-278); L1-resident, same for every block
    z[n1] = {reg[20 * n1 + q] * hw, reg[FROW + 20 * n1 + q] * hw};
  }
  dft20(z);
  __syncthreads();   // everyone has consumed the frame rows; region becomes U[k1][n2]
  float2* U = reinterpret_cast<float2*>(reg);
#pragma unroll
  for (int k1 = 0; k1 < 20; k1++) {
    const float2 t = twv[k1];
    cpx v = cmul(z[k1], t.x, t.y);
    WB_SETTLE2(v.re, v.im);
    U[k1 * UROW + q] = make_float2(v.re, v.im);
  }
  __syncthreads();
  // ---- stage 2: 20-point DFT over n2 for k1 = q: Z[q + 20 k2] ----
#pragma unroll
  for (int n2 = 0; n2 < 20; n2++) {
    float2 v = U[q * UROW + n2];
    z[n2] = {v.x, v.y};
  }
  dft20(z);
  __syncthreads();
  float2* Z = reinterpret_cast<float2*>(reg);   // Z[k], k = 0..399
#pragma unroll
  for (int k2 = 0; k2 < 20; k2++) { WB_SETTLE2(z[k2].re, z[k2].im); Z[q + 20 * k2] = make_float2(z[k2].re, z[k2].im); }
  __syncthreads();
  // ---- stage 3: split the packed transform, power spectrum of both frames ----
  // A[k] = (Z[k] + conj Z[400-k]) / 2,  B[k] = (Z[k] - conj Z[400-k]) / (2i)
  // this thread's 4 of the 80 x 16 filterbank taps (+ start / length of row tid): requested here, they land under
  // the unpack arithmetic and move to LDS next to the power spectra, so that stage 4's tap loop runs on LDS
  // latency instead of one dependent global load per tap
  const float4 tapq = *reinterpret_cast<const float4*>(&tabs->tap_w[tid * 4]);
  const int tap_s0 = tid < MEL_N_MELS ? tabs->tap_start[tid] : 0, tap_n = tid < MEL_N_MELS ? tabs->tap_len[tid] : 0;
  float pa[11], pb[11];
#pragma unroll
  for (int i = 0; i < 11; i++) {
    int k = q + 20 * i;
    if (k <= 200) {
      float2 zk = Z[k], zn = Z[k == 0 ? 0 : 400 - k];
      float ar = zk.x + zn.x, ai = zk.y - zn.y;
      float br = zk.y + zn.y, bi = zn.x - zk.x;
      pa[i] = 0.25f * (ar * ar + ai * ai);
      pb[i] = 0.25f * (br * br + bi * bi);
    }
  }
  __syncthreads();
  float* P = reg;   // P[0..200] frame a, P[209..409] frame b
#pragma unroll
  for (int i = 0; i < 11; i++) {
    int k = q + 20 * i;
    if (k <= 200) { WB_SETTLE2(pa[i], pb[i]); P[k] = pa[i]; P[PB_OFF + k] = pb[i]; }
  }
  {
    const int m = tid >> 2;                     // taps 4 tid .. 4 tid + 3 belong to mel row m
    *reinterpret_cast<float4*>(&lds[(m / 5) * 2 * FROW + TAP_OFF + (m % 5) * MEL_MAX_TAPS + (tid & 3) * 4]) = tapq;
    if (tid < MEL_N_MELS) {                     // (start, length) of row tid behind its pair region's taps
      int* meta = reinterpret_cast<int*>(&lds[(tid / 5) * 2 * FROW + TAP_OFF + 5 * MEL_MAX_TAPS + (tid % 5) * 2]);
      meta[0] = tap_s0; meta[1] = tap_n;
    }
  }
  __syncthreads();
  // ---- stage 4: sparse mel filterbank, log10, local max ----
  // lane = frame, half-wave = group of 8 mel rows: the filter taps are uniform over each half-wave
  // (broadcast LDS reads), the power spectra are the lane's own frame
  const float LOG10_2 = 0.30102999566398119521f;    // log10(2)
  float lmax = -INFINITY, lmin = INFINITY;
  {
    const int f = tid & (FPB - 1), grp = tid >> 5;            // 10 groups x 8 rows
    float* fr_reg = lds + (f >> 1) * 2 * FROW;
    const float* Pf = fr_reg + (f & 1) * PB_OFF;
    const bool live = f0 + f < w.n_frames;
    int s0v[8], lenv[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int m = grp * 8 + r;
      const int* meta = reinterpret_cast<const int*>(&lds[(m / 5) * 2 * FROW + TAP_OFF + 5 * MEL_MAX_TAPS + (m % 5) * 2]);
      s0v[r] = meta[0]; lenv[r] = meta[1];
    }
    // chunk-major: the eight rows of the group advance together (eight independent sums: their LDS reads are in flight
    // together -- a row-major loop waits for one row's chunk at a time: 1.09 -> 1.16 G frames/s, profiles/r04_h_mel_variants.txt); the trip count is the group's longest row,
    // taps past a row's length are stored as zeros and Pf[s0 + t] stays inside the pair's region (finite FFT leftovers)
    float accv[8];
    int nch = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) { accv[r] = 0.f; nch = max(nch, (lenv[r] + 3) >> 2); }
    for (int c = 0; c < nch; c++) {
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const int m = grp * 8 + r;
        const float* tw = lds + (m / 5) * 2 * FROW + TAP_OFF + (m % 5) * MEL_MAX_TAPS;
        const float4 w4 = *reinterpret_cast<const float4*>(tw + 4 * c);
        const float* pp = Pf + s0v[r] + 4 * c;
#if !defined(HIPEMU) && !defined(WB_MEL_NO_ASM)
        // plain v_fmac: left to itself the compiler SLP-packs the sums of two rows into v_pk_fma_f32 and builds each operand
        // pair with two or three v_mov (16 packed + ~40 moves per chunk against 32 scalar FMAs); same operation order
        asm("v_fmac_f32 %0, %1, %2" : "+v"(accv[r]) : "v"(w4.x), "v"(pp[0]));
        asm("v_fmac_f32 %0, %1, %2" : "+v"(accv[r]) : "v"(w4.y), "v"(pp[1]));
        asm("v_fmac_f32 %0, %1, %2" : "+v"(accv[r]) : "v"(w4.z), "v"(pp[2]));
        asm("v_fmac_f32 %0, %1, %2" : "+v"(accv[r]) : "v"(w4.w), "v"(pp[3]));
#else
        accv[r] += w4.x * pp[0]; accv[r] += w4.y * pp[1]; accv[r] += w4.z * pp[2]; accv[r] += w4.w * pp[3];
#endif
      }
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int m = grp * 8 + r;
      const float acc = accv[r];
      // tensor_max_scalar(x, 1e-10) = relu(x - 1e-10) + 1e-10 (helper.rs:8-10); log10 = ln/ln10 (:24-27)
      // ln(x) / ln 10 (helper.rs:24-27) through the hardware log2 (v_log_f32, <= 1 ulp; the argument is >= 1e-10, normal):
      // log10 x = log2 x * log10 2 -- within 2e-7 of the reference's two-step form, tolerance class "mel"
      const float v = __log2f(fmaxf(acc - 1.0e-10f, 0.f) + 1.0e-10f) * LOG10_2;
      // stored already normalised, (x + 4) / 4 (audio.rs:53); the clamp max(x, max - 8) (audio.rs:52) needs the window
      // maximum and is applied by the fix-up pass -- only to tiles whose minimum is below it (rare: 80 dB down)
      fr_reg[OT_OFF + 2 * m + (f & 1)] = (v + 4.0f) / 4.0f;
      if (live) lmax = fmaxf(lmax, v);
      if (f0 + f < w.n_emit) lmin = fminf(lmin, v);
    }
  }
  lmax = wave_max(lmax);
  lmin = -wave_max(-lmin);
  if ((tid & 63) == 0) { lds[BMAX_OFF + (tid >> 6)] = lmax; lds[BMAX_OFF + 8 + (tid >> 6)] = lmin; }   // free tail of pair 0's region
  __syncthreads();
  // one (max, min) pair per block, reduced by the fix-up pass: no atomics (235 of them per window would
  // serialise on one L2 word)
  if (tid == 0) {
    float bm = lds[BMAX_OFF], bn = lds[BMAX_OFF + 8];
#pragma unroll
    for (int i = 1; i < MEL_THREADS / 64; i++) { bm = fmaxf(bm, lds[BMAX_OFF + i]); bn = fminf(bn, lds[BMAX_OFF + 8 + i]); }
    gmax[((int64_t)blockIdx.y * bmax_stride + blockIdx.x) * 2] = bm;
    gmax[((int64_t)blockIdx.y * bmax_stride + blockIdx.x) * 2 + 1] = bn;
  }
  // ---- stage 5: coalesced store of the [80][32] tile ----
  float* o = out + (int64_t)blockIdx.y * win_stride + f0;
  const int nf = min(FPB, w.n_emit - f0);
  if (nf == FPB && (row_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
    // full tile: 80 rows x 8 pieces of 16 bytes, two per thread
#pragma unroll
    for (int i = 0; i < MEL_N_MELS * FPB / 4 / MEL_THREADS; i++) {
      const int e = tid + i * MEL_THREADS, m = e >> 3, f = (e & 7) * 4;
      const float* a = lds + (f >> 1) * 2 * FROW + OT_OFF + 2 * m;     // frames f, f+1 in one pair region, f+2, f+3 in the next
      *reinterpret_cast<float4*>(o + (int64_t)m * row_stride + f) = make_float4(a[0], a[1], a[2 * FROW], a[2 * FROW + 1]);
    }
  } else {
    for (int e = tid; e < MEL_N_MELS * FPB; e += MEL_THREADS) {
      int m = e / FPB, f = e - m * FPB;
      if (f < nf) o[(int64_t)m * row_stride + f] = lds[(f >> 1) * 2 * FROW + OT_OFF + 2 * m + (f & 1)];
      else if (f0 + f >= w.n_emit && f0 + f < zend) o[(int64_t)m * row_stride + f] = 0.f;
    }
  }
}

// Clamp fix-up (audio.rs:50-52): one block per (tile, window).  The window maximum is the maximum over the tiles'
// maxima; a tile whose minimum is not below max - 8 needs nothing (the main kernel already stored (x + 4) / 4) and
// its block exits after reading the per-tile (max, min) pairs -- for ordinary audio that is every tile, so the log-mel
// is written ONCE.  A tile that does dip below (digital silence, 80 dB under the loudest bin of the window) is
// rewritten: x < max - 8 gives (relu(x - m8) + m8 + 4) / 4 = (m8 + 4) / 4 exactly as the reference computes it.
__global__ __launch_bounds__(256) void mel_clamp_fixup_kernel(const MelWindow* __restrict__ wins, float* __restrict__ out,
                                                              int64_t win_stride, int row_stride,
                                                              const float* __restrict__ bmm, int bmax_stride) {
  __shared__ float wmax;
  const int blk = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
  const int nblk = (wins[w].n_frames + FPB - 1) / FPB;
  if (blk >= nblk) return;
  if (tid < 64) {   // global max of the window (audio.rs:50) = max over its tiles' maxima
    float v = -INFINITY;
    for (int i = tid; i < nblk; i += 64) v = fmaxf(v, bmm[((int64_t)w * bmax_stride + i) * 2]);
    v = wave_max(v);
    if (tid == 0) wmax = v;
  }
  __syncthreads();
  // audio.rs:50-53: max computed as f64 from the f32 max, (max - 8.0) handed back as f32
  const float m8 = (float)((double)wmax - 8.0);
  if (bmm[((int64_t)w * bmax_stride + blk) * 2 + 1] >= m8) return;      // nothing in this tile is clamped
  const float yfloor = (m8 + 4.0f) / 4.0f;                               // (relu(x - m8) + m8 + 4) / 4 for x < m8
  const int f0 = blk * FPB, nf = min(FPB, wins[w].n_emit - f0);
  float* o = out + (int64_t)w * win_stride + f0;
  for (int e = tid; e < MEL_N_MELS * FPB; e += 256) {
    const int m = e / FPB, f = e - m * FPB;
    if (f < nf) {
      float* q = o + (int64_t)m * row_stride + f;
      *q = fmaxf(*q, yfloor);
    }
  }
}

// 16-bit PCM -> f32, s / 32767 (the reference divides by 2^(b-1) - 1, bin/transcribe/main.rs:45-52); 8 samples per
// lane: one 16-byte load, two 16-byte stores; IEEE division, so equal to the host loop bit for bit
__global__ void pcm_s16_to_f32_kernel(const int16_t* __restrict__ src, int64_t n, float* __restrict__ dst) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i + 8 <= n && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    const uint4 v = *reinterpret_cast<const uint4*>(src + i);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    float o[8];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      o[2 * j] = __fdiv_rn((float)(int16_t)(w[j] & 0xffffu), 32767.0f);
      o[2 * j + 1] = __fdiv_rn((float)(int16_t)(w[j] >> 16), 32767.0f);
    }
    *reinterpret_cast<float4*>(dst + i) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(dst + i + 4) = make_float4(o[4], o[5], o[6], o[7]);
  } else {
    for (int64_t k = i; k < n && k < i + 8; k++) dst[k] = __fdiv_rn((float)src[k], 32767.0f);
  }
}

__global__ void fill_f32_kernel(float* p, int64_t n, float v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace

int mel_bmax_stride(int max_frames) { return (max_frames + FPB - 1) / FPB; }   // one (max, min) pair per tile with real frames

void launch_mel_spectrogram(hipStream_t st, const float* pcm, const MelWindow* wins_dev, int n_windows,
                            int max_frames, const MelTables* tabs_dev, float* out, int64_t win_stride,
                            int row_stride, float* bmax_dev, int pad, int pad_limit) {
  // tiles up to the last frame OR the last zero-padding frame of the longest window
  dim3 grid((std::min(max_frames + pad, std::max(pad_limit, max_frames)) + FPB - 1) / FPB, n_windows);
  hipLaunchKernelGGL(mel_spectrogram_kernel, grid, dim3(MEL_THREADS), 0, st, pcm, wins_dev, tabs_dev, out,
                     win_stride, row_stride, bmax_dev, mel_bmax_stride(max_frames), pad, pad_limit);
}

void launch_mel_finalize(hipStream_t st, const MelWindow* wins_dev, int n_windows, float* out, int64_t win_stride,
                         int row_stride, const float* bmax_dev, int max_frames) {
  dim3 grid((max_frames + FPB - 1) / FPB, n_windows);
  hipLaunchKernelGGL(mel_clamp_fixup_kernel, grid, dim3(256), 0, st, wins_dev, out, win_stride, row_stride, bmax_dev,
                     mel_bmax_stride(max_frames));
}

void launch_pcm_s16_to_f32(hipStream_t st, const int16_t* src, int64_t n, float* dst) {
  if (n <= 0) return;
  const int64_t threads = (n + 7) / 8;
  hipLaunchKernelGGL(pcm_s16_to_f32_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, src, n, dst);
}

void launch_fill_f32(hipStream_t st, float* p, int64_t n, float v) {
  if (n <= 0) return;
  hipLaunchKernelGGL(fill_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n, v);
}

// ---- host: constant tables ------------------------------------------------------------
// The filterbank follows the f32 op order of audio.rs:67-143 / :178-266 (scalars are cast to
// f32 before each tensor op, as burn-tch does), so it matches what the reference builds on
// device for every window -- here once per process.
int mel_tables_build(double sample_rate, MelTables* t) {
  memset(t, 0, sizeof(*t));
  // Hann: sin(n * f32(pi/400))^2 in f32 (audio.rs:272-278)
  const float step = (float)(M_PI / 400.0);
  for (int n = 0; n < 400; n++) {
    float s = sinf((float)n * step);
    t->hann[n] = s * s;
  }
  for (int n2 = 0; n2 < 20; n2++)
    for (int k1 = 0; k1 < 20; k1++) {
      double a = -2.0 * M_PI * (double)(n2 * k1) / 400.0;
      t->tw[n2 * 20 + k1] = make_float2((float)cos(a), (float)sin(a));
    }
  // mel_frequencies_device (audio.rs:178-196) with hz_to_mel (:198-230) in f64 on the host
  auto hz_to_mel = [](double f) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp;
  };
  const int n_mels = 80, n_f = n_mels + 2;
  const double fmin = 0.0, fmax = sample_rate * 0.5;
  const double min_mel = hz_to_mel(fmin), max_mel = hz_to_mel(fmax);
  const float mstep = (float)((max_mel - min_mel) / (double)(n_f - 1)), mmin = (float)min_mel;
  const float f_sp = (float)(200.0 / 3.0), min_log_mel = (float)(1000.0 / (200.0 / 3.0));
  const float logstep = (float)(log(6.4) / 27.0), min_log_hz = 1000.0f;
  float mel_f[82];
  for (int i = 0; i < n_f; i++) {
    float mel = (float)i * mstep + mmin;
    // mel_to_hz_tensor (audio.rs:232-266): log_t * (exp((mel - mlm) * logstep) * 1000) + (1 - log_t) * (mel * f_sp + 0)
    float log_t = mel >= min_log_mel ? 1.0f : 0.0f;
    float a = log_t * (expf((mel - min_log_mel) * logstep) * min_log_hz);
    float b = (-log_t + 1.0f) * (mel * f_sp + 0.0f);
    mel_f[i] = a + b;
  }
  const float fstep = (float)(sample_rate / 400.0);
  float W[80][201];
  for (int i = 0; i < n_mels; i++) {
    float fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
    float enorm = powf(mel_f[i + 2] - mel_f[i], -1.0f) * 2.0f;
    for (int j = 0; j < 201; j++) {
      float ff = (float)j * fstep;
      float lower = -(mel_f[i] - ff) / fd0;
      float upper = (mel_f[i + 2] - ff) / fd1;
      // tensor_min(lower, upper) = -(relu(-lower - (-upper)) + (-upper))  (helper.rs:16-22)
      float nx = -lower, nm = -upper;
      float tmin = -(fmaxf(nx - nm, 0.f) + nm);
      W[i][j] = fmaxf(tmin, 0.f) * enorm;
    }
  }
  for (int i = 0; i < n_mels; i++) {
    int s = -1, e = -1;
    for (int j = 0; j < 201; j++)
      if (W[i][j] != 0.f) { if (s < 0) s = j; e = j; }
    if (s < 0) { s = 0; e = -1; }
    int len = e - s + 1;
    if (len > MEL_MAX_TAPS) return -1;
    t->tap_start[i] = s;
    t->tap_len[i] = len;
    for (int k = 0; k < len; k++) t->tap_w[i * MEL_MAX_TAPS + k] = W[i][s + k];
  }
  return 0;
}

}  // namespace wb
