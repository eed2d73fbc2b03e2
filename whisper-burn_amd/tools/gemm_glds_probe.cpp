// Developer probe (round 6): the split-precision GEMM with BOTH operands as fp16 pieces and a direct-to-LDS operand path
// (global_load_lds_dwordx4, counted vmcnt, raw barriers, a ring of LDS stages) against the product kernel
// (gemm_f16x3_kernel<128, 128, 2, 2, 1, APRE = true>) at large-v2's / small's encoder shapes.  Same MFMA order over k ->
// outputs must be bit-identical.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip tools/gemm_glds_probe.cpp -o tools/gemm_glds_probe
#include "../csrc/gemm_f16x3.hip"
#include <cstdio>
#include <cstring>
#include <vector>
namespace wb { void prof_tag(int, double) {} bool prof_take_events(hipEvent_t*, hipEvent_t*) { return false; } }
using namespace wb;

namespace {
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
// one 16-byte piece per lane, global -> LDS at (wave-uniform byte address) + 16 lane; absent from hipcc's waitcnt bookkeeping
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory"); }

// 128 x 128 tile, 4 waves (2 x 2, 64 x 64 each), BK = 32.  A stage = four 8 KB piece tiles [128 rows][4 slots of 16 B],
// physical slot = logical slot ^ ((row >> 2) & 3): conflict-free ds_read_b128 for the MFMA operand pattern (lanes walk rows).
template <int STAGES>
__global__ __launch_bounds__(256, STAGES <= 2 ? 2 : 1) void gemm_glds_kernel(GemmArgs g, const u16* __restrict__ Wh,
                                                                             const u16* __restrict__ Wl, int ldwt) {
  constexpr int BM = 128, BN = 128, PIECE = 128 * 32;          // halves per piece tile
  __shared__ __attribute__((aligned(16))) u16 smem[STAGES][4][PIECE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int M = g.M, N = g.N, K = g.K;
  const int nk = K / 32;
  // this thread's two LDS items per piece: p = i * 256 + tid -> row p >> 2, physical slot p & 3
  const u16* srcA_h[2]; const u16* srcA_l[2]; const u16* srcB_h[2]; const u16* srcB_l[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int p = i * 256 + tid, row = p >> 2, ps = p & 3, ls = ps ^ ((row >> 2) & 3);
    const int ma = min(m0 + row, M - 1), nb = min(n0 + row, N - 1);      // (rows past the edge re-read the last row: discarded)
    srcA_h[i] = g.Ah + (int64_t)ma * g.lda + ls * 8; srcA_l[i] = g.Al + (int64_t)ma * g.lda + ls * 8;
    srcB_h[i] = Wh + (int64_t)nb * ldwt + ls * 8; srcB_l[i] = Wl + (int64_t)nb * ldwt + ls * 8;
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto issue = [&](int t, int stage) {
    const int k0 = t * 32;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const unsigned off = (unsigned)((i * 256 + wave_u * 64) * 16);
      glds16(srcA_h[i] + k0, lds_addr(&smem[stage][0][0]) + off);
      glds16(srcA_l[i] + k0, lds_addr(&smem[stage][1][0]) + off);
      glds16(srcB_h[i] + k0, lds_addr(&smem[stage][2][0]) + off);
      glds16(srcB_l[i] + k0, lds_addr(&smem[stage][3][0]) + off);
    }
  };
  f32x16 acc[2][2], acl[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) { acc[i][j][r] = 0.f; acl[i][j][r] = 0.f; }
  auto frag = [&](int stage, int piece, int row, int ls) -> f16x8 {
    return *reinterpret_cast<const f16x8*>(&smem[stage][piece][(row * 4 + (ls ^ ((row >> 2) & 3))) * 8]);
  };
  auto compute = [&](int stage) {
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int row = wm * 64 + i * 32 + li;
        ah[i] = frag(stage, 0, row, ks * 2 + lh); al[i] = frag(stage, 1, row, ks * 2 + lh);
      }
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int row = wn * 64 + j * 32 + li;
        bh[j] = frag(stage, 2, row, ks * 2 + lh); bl[j] = frag(stage, 3, row, ks * 2 + lh);
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          acl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acl[i][j], 0, 0, 0);
          acl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acl[i][j], 0, 0, 0);
        }
    }
  };
  // ring: tiles t + 1 .. t + STAGES - 1 in flight while tile t is consumed; 8 LDS-DMA instructions per wave and tile
  constexpr int AHEAD = STAGES - 1;
#pragma unroll
  for (int s = 0; s < AHEAD; s++)
    if (s < nk) issue(s, s);
  for (int t = 0; t < nk; t++) {
    // tile t has landed once at most the later tiles' instructions are outstanding
    const int later = min(nk - 1 - t, AHEAD - 1);
    if (later >= 3) wait_vm<24>(); else if (later == 2) wait_vm<16>(); else if (later == 1) wait_vm<8>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();                    // every wave's pieces of tile t are in LDS; everyone is done with tile t - 1
    if (t + AHEAD < nk) issue(t + AHEAD, (t + AHEAD) % STAGES);      // ... whose stage is refilled now
    compute(t % STAGES);
  }
  // epilogue: bias only (the probe compares C)
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int col = n0 + wn * 64 + j * 32 + li;
      const bool col_ok = col < N;
      const float bias = g.bias ? g.bias[col_ok ? col : N - 1] : 0.f;
      const int rbase = m0 + wm * 64 + i * 32 + 4 * lh;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        const float v = (acc[i][j][r] + acl[i][j][r] * LO_UNSCALE) + bias;
        if (col_ok && row < M) g.C[(int64_t)row * g.ldc + col] = v;
      }
    }
}

// The same ring (2 stages, 2 blocks per CU) with the k-loop SKEWED so that the LDS reads of the next k-step are always issued
// before the 12 MFMAs of the current one (the plain loop above exposes two LDS round trips per k-tile: ISA schedule
// "8 ds_read, wait, 12 MFMA, 8 ds_read, wait, 12 MFMA"):
//   [read F(t, ks1)] [12 MFMA on F(t, ks0)] [tile t+1 landed? ; barrier] [refill stage of tile t with t+2] [read F(t+1, ks0)] [12 MFMA on F(t, ks1)]
__global__ __launch_bounds__(256, 2) void gemm_glds_pipe_kernel(GemmArgs g, const u16* __restrict__ Wh, const u16* __restrict__ Wl,
                                                                 int ldwt) {
  constexpr int BM = 128, BN = 128, PIECE = 128 * 32, STAGES = 2;
  __shared__ __attribute__((aligned(16))) u16 smem[STAGES][4][PIECE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int M = g.M, N = g.N, K = g.K;
  const int nk = K / 32;
  const u16* srcA_h[2]; const u16* srcA_l[2]; const u16* srcB_h[2]; const u16* srcB_l[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int p = i * 256 + tid, row = p >> 2, ps = p & 3, ls = ps ^ ((row >> 2) & 3);
    const int ma = min(m0 + row, M - 1), nb = min(n0 + row, N - 1);
    srcA_h[i] = g.Ah + (int64_t)ma * g.lda + ls * 8; srcA_l[i] = g.Al + (int64_t)ma * g.lda + ls * 8;
    srcB_h[i] = Wh + (int64_t)nb * ldwt + ls * 8; srcB_l[i] = Wl + (int64_t)nb * ldwt + ls * 8;
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto issue = [&](int t, int stage) {
    const int k0 = t * 32;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const unsigned off = (unsigned)((i * 256 + wave_u * 64) * 16);
      glds16(srcA_h[i] + k0, lds_addr(&smem[stage][0][0]) + off);
      glds16(srcA_l[i] + k0, lds_addr(&smem[stage][1][0]) + off);
      glds16(srcB_h[i] + k0, lds_addr(&smem[stage][2][0]) + off);
      glds16(srcB_l[i] + k0, lds_addr(&smem[stage][3][0]) + off);
    }
  };
  f32x16 acc[2][2], acl[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) { acc[i][j][r] = 0.f; acl[i][j][r] = 0.f; }
  struct Frags { f16x8 ah[2], al[2], bh[2], bl[2]; };
  auto read_frags = [&](Frags& f, int stage, int ks) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int row = wm * 64 + i * 32 + li, sl = (ks * 2 + lh) ^ ((row >> 2) & 3);
      f.ah[i] = *reinterpret_cast<const f16x8*>(&smem[stage][0][(row * 4 + sl) * 8]);
      f.al[i] = *reinterpret_cast<const f16x8*>(&smem[stage][1][(row * 4 + sl) * 8]);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int row = wn * 64 + j * 32 + li, sl = (ks * 2 + lh) ^ ((row >> 2) & 3);
      f.bh[j] = *reinterpret_cast<const f16x8*>(&smem[stage][2][(row * 4 + sl) * 8]);
      f.bl[j] = *reinterpret_cast<const f16x8*>(&smem[stage][3][(row * 4 + sl) * 8]);
    }
  };
  // half a k-step: the six MFMAs of A row block i (the accumulation order per accumulator is the plain loop's)
  auto mma_half = [&](const Frags& f, int i) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
      acl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bl[j], acl[i][j], 0, 0, 0);
      acl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bh[j], acl[i][j], 0, 0, 0);
    }
  };
  Frags f0, f1;
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  if (nk > 1) wait_vm<8>(); else wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  read_frags(f0, 0, 0);
  for (int t = 0; t < nk; t++) {
    const int stage = t & 1;
    // the reads of a k-step sit BETWEEN the two MFMA halves of the step before it: never in front of a wait
    mma_half(f0, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_frags(f1, stage, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma_half(f0, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < nk) {
      // tile t + 1 has landed for this wave (its 8 instructions are the only ones outstanding) and this wave's reads of
      // tile t are complete (issued six MFMAs ago): behind the barrier both hold for every wave -> stage t % 2 is free
      wait_vm<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + 2 < nk) issue(t + 2, stage);
    }
    __builtin_amdgcn_sched_barrier(0);
    mma_half(f1, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < nk) read_frags(f0, stage ^ 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma_half(f1, 1);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int col = n0 + wn * 64 + j * 32 + li;
      const bool col_ok = col < N;
      const float bias = g.bias ? g.bias[col_ok ? col : N - 1] : 0.f;
      const int rbase = m0 + wm * 64 + i * 32 + 4 * lh;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        const float v = (acc[i][j][r] + acl[i][j][r] * LO_UNSCALE) + bias;
        if (col_ok && row < M) g.C[(int64_t)row * g.ldc + col] = v;
      }
    }
}
__global__ void split_rows_kernel(const float* x, u16* hi, u16* lo, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { u16 h, l; split1(x[i], h, l); hi[i] = h; lo[i] = l; }
}
}  // namespace

int main() {
  struct Shape { const char* name; int M, K, N; };
  const Shape shapes[] = {{"large qkv", 28500, 1280, 3840}, {"large out", 28500, 1280, 1280}, {"large mlp1", 28500, 1280, 5120},
                          {"large mlp2", 28500, 1024, 1280}, {"small qkv", 38250, 768, 2304}, {"small mlp1", 38250, 768, 3072},
                          {"tiny qkv", 1868, 384, 1152}};
  size_t maxA = 0, maxB = 0, maxC = 0;
  for (auto& s : shapes) { maxA = std::max(maxA, (size_t)s.M * s.K); maxB = std::max(maxB, (size_t)s.K * s.N); maxC = std::max(maxC, (size_t)s.M * s.N); }
  float *A, *B, *C, *bias; u16 *Ah, *Al, *Wh, *Wl;
  hipMalloc(&A, maxA * 4); hipMalloc(&B, maxB * 4); hipMalloc(&C, maxC * 4); hipMalloc(&bias, 8192 * 4);
  hipMalloc(&Ah, maxA * 2); hipMalloc(&Al, maxA * 2); hipMalloc(&Wh, maxB * 2 + 4096); hipMalloc(&Wl, maxB * 2 + 4096);
  std::vector<float> h(std::max(maxA, maxB));
  for (size_t i = 0; i < h.size(); i++) h[i] = (float)((int)((i * 2654435761u) >> 13 & 4095) - 2048) * 0.0007f;
  hipMemcpy(A, h.data(), maxA * 4, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), maxB * 4, hipMemcpyHostToDevice);
  hipMemcpy(bias, h.data(), 8192 * 4, hipMemcpyHostToDevice);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ref(maxC), got(maxC);
  const int REP = 10;
  for (auto& s : shapes) {
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)(((size_t)s.M * s.K + 255) / 256)), dim3(256), 0, st, A, Ah, Al, (int64_t)s.M * s.K);
    launch_split_weight_f16(st, B, s.K, s.N, Wh, Wl);
    GemmArgs g;
    g.A = A; g.lda = s.K; g.Ah = Ah; g.Al = Al; g.C = C; g.ldc = s.N; g.bias = bias; g.M = s.M; g.N = s.N; g.K = s.K;
    const double flop = 2.0 * s.M * s.K * s.N;
    const dim3 grid((s.N + 127) / 128, (s.M + 127) / 128);
    for (int cfg = 0; cfg < 5; cfg++) {
      auto run = [&]() {
        if (cfg == 0) hipLaunchKernelGGL((gemm_f16x3_kernel<128, 128, 2, 2, 1, true>), grid, dim3(256), 0, st, g, Wh, Wl, s.K);
        else if (cfg == 1) hipLaunchKernelGGL((gemm_glds_kernel<2>), grid, dim3(256), 0, st, g, Wh, Wl, s.K);
        else if (cfg == 2) hipLaunchKernelGGL((gemm_glds_kernel<3>), grid, dim3(256), 0, st, g, Wh, Wl, s.K);
        else if (cfg == 3) hipLaunchKernelGGL((gemm_glds_kernel<4>), grid, dim3(256), 0, st, g, Wh, Wl, s.K);
        else hipLaunchKernelGGL(gemm_glds_pipe_kernel, grid, dim3(256), 0, st, g, Wh, Wl, s.K);
      };
      hipMemsetAsync(C, 0, (size_t)s.M * s.N * 4, st);
      run();
      if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { printf("%s cfg %d: launch failed\n", s.name, cfg); continue; }
      hipMemcpy(got.data(), C, (size_t)s.M * s.N * 4, hipMemcpyDeviceToHost);
      if (cfg == 0) ref = got;
      size_t ndiff = 0; double maxd = 0;
      for (size_t i = 0; i < (size_t)s.M * s.N; i++) if (ref[i] != got[i]) { ndiff++; maxd = std::max(maxd, (double)fabsf(ref[i] - got[i])); }
      for (int i = 0; i < 3; i++) run();
      hipEventRecord(e0, st);
      for (int i = 0; i < REP; i++) run();
      hipEventRecord(e1, st);
      hipStreamSynchronize(st);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%-11s M=%5d K=%4d N=%4d %-18s %8.1f us  %6.1f TF/s (f32-equivalent)  %s (%zu differ, max %.3g)\n", s.name, s.M, s.K, s.N,
             cfg == 0 ? "product APRE" : cfg == 1 ? "glds 2 stages" : cfg == 2 ? "glds 3 stages" : cfg == 3 ? "glds 4 stages" : "glds 2 st. skewed", ms * 1e3 / REP,
             flop / (ms * 1e-3 / REP) / 1e12, ndiff == 0 ? "bit-identical" : "DIFFERS", ndiff, maxd);
    }
  }
  return 0;
}
