// Developer probe (not part of the library): how fast can ONE launch pull a 26 MB weight matrix (K = 1280 rows of N = 5120
// floats: large-v2's first MLP product) when every wave requests all of its bytes up front, as the skinny GEMM of batch mode
// does -- and does the SHAPE of a wave's requests matter?  The skinny GEMM reads 256-byte row pieces (16 lanes x 16 B, four
// K-rows per load instruction) and takes 19.5 us for this matrix (profiles/r03_m_layer_cycle_large_v2.txt).
//
//   hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.cpp && ./stream_probe
//
// 2560 waves (320 blocks x 8), 10 float4 loads per lane each, always the whole matrix:
//   piece 256 / 512 / 1024   a wave's load instruction covers 4 / 2 / 1 K-rows x 256 / 512 / 1024 contiguous bytes
//   linear                   a wave reads 10 KB of consecutive bytes
// 12 matrices (315 MB) are read in rotation so that neither L2 nor the 256 MB memory-side cache serves a repeat.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int K = 1280, N = 5120, NLD = 10, NMAT = 12;

template <int PIECE>   // bytes of one row piece; 0 = linear
__global__ __launch_bounds__(512) void read_kernel(const float* __restrict__ W, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 v[NLD];
  if constexpr (PIECE == 0) {
    const int w = blockIdx.x * 8 + wave;                                 // 2560 waves x 10 KB
    const float* p = W + (size_t)w * (NLD * 256) + lane * 4;
#pragma unroll
    for (int t = 0; t < NLD; t++) v[t] = *reinterpret_cast<const float4*>(p + t * 256);
  } else {
    constexpr int LPR = PIECE / 16, RPI = 64 / LPR;                      // lanes per row piece, rows per instruction
    constexpr int STRIPS = N * 4 / PIECE, ROWS_PER_WAVE = NLD * RPI;
    const int strip = blockIdx.x % STRIPS, grp = (blockIdx.x / STRIPS) * 8 + wave;
    const float* p = W + (size_t)(grp * ROWS_PER_WAVE + lane / LPR) * N + strip * (PIECE / 4) + (lane % LPR) * 4;
#pragma unroll
    for (int t = 0; t < NLD; t++) v[t] = *reinterpret_cast<const float4*>(p + (size_t)(t * RPI) * N);
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NLD; t++) s += v[t].x + v[t].y + v[t].z + v[t].w;
  if (s == 12345.678f) sink[0] = s;
}

template <int PIECE>
static void run(const char* name, const float* W, float* sink) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  const int reps = 240;
  for (int i = 0; i < NMAT; i++) hipLaunchKernelGGL(read_kernel<PIECE>, dim3(320), dim3(512), 0, 0, W + (size_t)i * K * N, sink);
  CHECK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(read_kernel<PIECE>, dim3(320), dim3(512), 0, 0, W + (size_t)(i % NMAT) * K * N, sink);
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, a, b));
  // back-to-back independent launches overlap their ramps: this is the sustained rate; the dependent-launch cost is in
  // profiles/r03_m_layer_cycle_large_v2.txt
  printf("%-10s %7.2f us per 26.2 MB launch (back to back)  = %5.2f TB/s\n", name, ms * 1e3f / reps, 26.2144e6 / (ms * 1e-3 / reps) / 1e12);
  // one launch at a time (synchronised): ramp + latency + stream + drain
  float tot = 0.f;
  for (int i = 0; i < 40; i++) {
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(read_kernel<PIECE>, dim3(320), dim3(512), 0, 0, W + (size_t)((i + 5) % NMAT) * K * N, sink);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    CHECK(hipEventElapsedTime(&ms, a, b));
    tot += ms;
  }
  printf("%-10s %7.2f us per launch, one at a time (event to event)\n", name, tot * 1e3f / 40);
}

int main() {
  float* W; float* sink;
  CHECK(hipMalloc(&W, (size_t)NMAT * K * N * 4));
  CHECK(hipMemset(W, 0, (size_t)NMAT * K * N * 4));
  CHECK(hipMalloc(&sink, 4));
  run<256>("piece 256", W, sink);
  run<512>("piece 512", W, sink);
  run<1024>("piece 1024", W, sink);
  run<0>("linear", W, sink);
  return 0;
}
