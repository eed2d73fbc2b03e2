// Developer probe (GPU, standalone): do two kernels with different static LDS sizes, running at the same time from two streams,
// ever see each other's LDS?  Kernel A (256 threads, LDS_A bytes) and kernel B (320 threads, LDS_B bytes) each fill their whole
// LDS allocation with a block-private pattern, synchronise, spin, and verify it; mismatches are logged with their byte offset.
//   hipcc --offload-arch=gfx950 -O2 -o lds_concurrency_probe lds_concurrency_probe.cpp && ./lds_concurrency_probe [seconds=2]
#include <hip/hip_runtime.h>
#if defined(WITH_LIB)
#include "kernels.h"   // csrc/: the library's own launchers (link with libwhisper_hip.so)
#endif

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Log { unsigned n; unsigned off[4096]; unsigned got[4096]; unsigned want[4096]; };

template <int BYTES, int NT, unsigned TAG>
__global__ __launch_bounds__(NT) void lds_kernel(Log* log, int iters, unsigned launch) {
  __shared__ __attribute__((aligned(16))) unsigned lds[BYTES / 4];
  const unsigned base = TAG | ((launch & 0xff) << 16) | (blockIdx.x & 0xffff);
  for (int it = 0; it < iters; it++) {
    const unsigned pat = base ^ ((unsigned)it << 24);
    for (int i = threadIdx.x; i < BYTES / 4; i += NT) lds[i] = pat + i * 0x01000193u;
    __syncthreads();
    // (a little arithmetic so the block stays resident while others start next to it)
    float acc = 0.f;
    for (int k = 0; k < 64; k++) acc = acc * 1.0001f + (float)k;
    if (acc == 12345.f) lds[0] = 0;
    __syncthreads();
    // verify with a different thread -> word mapping than the fill
    for (int i = NT - 1 - threadIdx.x; i < BYTES / 4; i += NT) {
      const unsigned want = pat + i * 0x01000193u, got = lds[i];
      if (got != want) {
        const unsigned slot = atomicAdd(&log->n, 1u);
        if (slot < 4096) { log->off[slot] = i * 4; log->got[slot] = got; log->want[slot] = want; }
      }
    }
    __syncthreads();
  }
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
  Log *la, *lb;
  CK(hipHostMalloc((void**)&la, sizeof(Log), hipHostMallocMapped));
  CK(hipHostMalloc((void**)&lb, sizeof(Log), hipHostMallocMapped));
  la->n = lb->n = 0;
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
#if defined(WITH_LIB)
  // aggressor = the library's split-precision GEMM (64 x 64 tiles, 40 960 B of LDS) instead of kernel A
  const int M = 2823, N = 384, K = 128;
  float *A, *Cm; unsigned short *Wh, *Wl;
  CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&Cm, (size_t)M * N * 4));
  CK(hipMalloc(&Wh, (size_t)N * K * 2)); CK(hipMalloc(&Wl, (size_t)N * K * 2));
  CK(hipMemset(A, 0, (size_t)M * K * 4)); CK(hipMemset(Wh, 0, (size_t)N * K * 2)); CK(hipMemset(Wl, 0, (size_t)N * K * 2));
  wb::GemmArgs g;
  g.A = A; g.lda = K; g.C = Cm; g.ldc = N; g.M = M; g.N = N; g.K = K;
  const bool f32 = argc > 2 && argv[2][0] == 'f';
  float* Bf = nullptr;
  if (f32) { CK(hipMalloc(&Bf, (size_t)K * N * 4)); CK(hipMemset(Bf, 0, (size_t)K * N * 4)); g.B = Bf; g.ldb = N; }
#endif
  const auto t0 = std::chrono::steady_clock::now();
  unsigned launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    for (int i = 0; i < 8; i++) {
#if defined(WITH_LIB)
      if (f32) wb::launch_gemm_f32(sa, g); else wb::launch_gemm_f16x3(sa, g, Wh, Wl, K);
#else
      hipLaunchKernelGGL((lds_kernel<40960, 256, 0xA0000000u>), dim3(176), dim3(256), 0, sa, la, 4, launches);
#endif
      hipLaunchKernelGGL((lds_kernel<54528, 320, 0xB0000000u>), dim3(188), dim3(320), 0, sb, lb, 4, launches);
      launches++;
    }
    CK(hipStreamSynchronize(sa));
    CK(hipStreamSynchronize(sb));
  }
  CK(hipDeviceSynchronize());
  printf("%u launch pairs; kernel A (40960 B, 256 threads): %u bad words; kernel B (54528 B, 320 threads): %u bad words\n",
         launches, la->n, lb->n);
  for (Log* l : {la, lb}) {
    std::map<unsigned, unsigned> hist;
    const unsigned n = l->n < 4096 ? l->n : 4096;
    for (unsigned i = 0; i < n; i++) hist[l->off[i] / 256 * 256]++;
    printf("  %s: offsets (256-byte bins):", l == la ? "A" : "B");
    int shown = 0;
    for (auto& kv : hist) if (shown++ < 40) printf(" %u:%u", kv.first, kv.second);
    printf("\n");
    for (unsigned i = 0; i < n && i < 6; i++) printf("    off %u got %08x want %08x\n", l->off[i], l->got[i], l->want[i]);
  }
  return 0;
}
