// Developer probe (GPU, standalone): does a kernel made of packed-FP32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32 -- the 20-point DFT butterflies of mel.hip) compute the same bits when an MFMA kernel runs next to it from
// another stream?  Victim: every thread runs dft20 R times on a thread-private sequence and stores a checksum; the result of a
// quiet run is the reference.  Aggressor (argv[2]): 0 = v_mfma_f32_32x32x16_f16 loop, 1 = v_mfma_f32_32x32x2_f32 loop,
// 2 = v_mfma_f32_16x16x32_f16 loop, 3 = plain VALU loop, 4 = none; 10 / 11 / 12 = as 0 with a register allocation of 152 / 144 / 136
// VGPRs, 13 = as 3 with 152; 14 / 15 = as 1 / 2 with 152; 16 / 17 / 18 = as 0 with 128 / 120 / 96;
// 30 / 31 / 32 = aggressor 15 against a victim whose own allocation is 152 / 128 / 96 VGPRs (default victim: 64);
// 40 = the 152-VGPR f16 aggressor and the 152-VGPR victim back to back on ONE stream.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../csrc -o pk_mfma_probe pk_mfma_probe.cpp && ./pk_mfma_probe [seconds] [aggressor]
#include "../csrc/mel.hip"   // (the anonymous-namespace helpers: dft20, cpx)

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

using namespace wb;

template <int TOPV>
__global__ __launch_bounds__(320, 2) void victim_kernel_t(float* out, int reps) {
  // (kinds 30 - 32: the victim with a register allocation of its own choosing: 152 / 128 / 96)
  if constexpr (TOPV == 151) asm volatile("v_mov_b32 v151, 0" ::: "v151");
  if constexpr (TOPV == 127) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  if constexpr (TOPV == 95) asm volatile("v_mov_b32 v95, 0" ::: "v95");
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  cpx z[20];
#pragma unroll
  for (int i = 0; i < 20; i++) z[i] = cpx{(float)((gid * 7 + i * 13) % 97) * 0.01f - 0.4f, (float)((gid * 5 + i * 11) % 89) * 0.01f - 0.3f};
  for (int r = 0; r < reps; r++) {
    dft20(z);
#pragma unroll
    for (int i = 0; i < 20; i++) { z[i].re *= 0.2236068f; z[i].im *= 0.2236068f; }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 20; i++) s += z[i].re * (float)(i + 1) + z[i].im * (float)(21 + i);
  out[gid] = s;
}

__global__ __launch_bounds__(320, 4) void victim_kernel(float* out, int reps) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  cpx z[20];
#pragma unroll
  for (int i = 0; i < 20; i++) z[i] = cpx{(float)((gid * 7 + i * 13) % 97) * 0.01f - 0.4f, (float)((gid * 5 + i * 11) % 89) * 0.01f - 0.3f};
  for (int r = 0; r < reps; r++) {
    dft20(z);
#pragma unroll
    for (int i = 0; i < 20; i++) { z[i].re *= 0.2236068f; z[i].im *= 0.2236068f; }   // 1 / sqrt(20): the sequence stays bounded
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 20; i++) s += z[i].re * (float)(i + 1) + z[i].im * (float)(21 + i);
  out[gid] = s;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// (TOP: the highest VGPR the kernel touches -- its register allocation is TOP + 1 rounded up to 8; the 64 x 64 split-precision
// GEMM allocates 152, which puts the NEXT wave on the SIMD at a register base that is not a multiple of 16)
template <int KIND, int TOP>
__global__ __launch_bounds__(256, 2) void aggressor_kernel(float* sink, int iters) {
  __shared__ float pad[40960 / 4];     // the LDS footprint of the 64 x 64 split-precision GEMM
  if constexpr (TOP == 151) asm volatile("v_mov_b32 v151, 0" ::: "v151");
  if constexpr (TOP == 143) asm volatile("v_mov_b32 v143, 0" ::: "v143");
  if constexpr (TOP == 135) asm volatile("v_mov_b32 v135, 0" ::: "v135");
  if constexpr (TOP == 127) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  if constexpr (TOP == 119) asm volatile("v_mov_b32 v119, 0" ::: "v119");
  if constexpr (TOP == 95) asm volatile("v_mov_b32 v95, 0" ::: "v95");
  pad[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  float keep = pad[(threadIdx.x * 7) & 255];
  if constexpr (KIND == 0) {
    f32x16 acc[4] = {};
    f16x8 a = {}, b = {};
    a[0] = (_Float16)keep;
    for (int i = 0; i < iters; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    keep += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  } else if constexpr (KIND == 1) {
    f32x16 acc[4] = {};
    for (int i = 0; i < iters; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(keep, 0.f, acc[j], 0, 0, 0);
    keep += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  } else if constexpr (KIND == 2) {
    f32x4 acc[4] = {};
    f16x8 a = {}, b = {};
    a[0] = (_Float16)keep;
    for (int i = 0; i < iters; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
    keep += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  } else {
    float x = keep;
    for (int i = 0; i < iters * 16; i++) x = x * 1.0001f + 0.5f;
    keep += x;
  }
  if (keep == 123456.789f) sink[0] = keep;
}

// ONE kernel whose even blocks run the f16 MFMA loop and whose odd blocks run the packed-FP32 butterflies (same register
// allocation for both, as inside a GEMM whose blocks are in different phases): kinds 20 (natural allocation) / 21 (152 VGPRs)
template <int TOP>
__global__ __launch_bounds__(320, 2) void mixed_kernel(float* out, int reps, float* sink, int iters) {
  if constexpr (TOP == 151) asm volatile("v_mov_b32 v151, 0" ::: "v151");
  if (blockIdx.x & 1) {
    const int gid = (blockIdx.x >> 1) * blockDim.x + threadIdx.x;
    cpx z[20];
#pragma unroll
    for (int i = 0; i < 20; i++) z[i] = cpx{(float)((gid * 7 + i * 13) % 97) * 0.01f - 0.4f, (float)((gid * 5 + i * 11) % 89) * 0.01f - 0.3f};
    for (int r = 0; r < reps; r++) {
      dft20(z);
#pragma unroll
      for (int i = 0; i < 20; i++) { z[i].re *= 0.2236068f; z[i].im *= 0.2236068f; }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 20; i++) s += z[i].re * (float)(i + 1) + z[i].im * (float)(21 + i);
    out[gid] = s;
  } else {
    f32x4 acc[4] = {};
    f16x8 a = {}, b = {};
    a[0] = (_Float16)(float)threadIdx.x;
    for (int i = 0; i < iters; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
    const float keep = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (keep == 123456.789f) sink[0] = keep;
  }
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
  const int kind = argc > 2 ? atoi(argv[2]) : 0;
  const int blocks = 188 * 4, n = blocks * 320, reps = 6;
  float *out, *sink;
  CK(hipMalloc(&out, (size_t)n * 4)); CK(hipMalloc(&sink, 4));
  hipStream_t sv, sa;
  CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  std::vector<float> ref(n), got(n);
  hipLaunchKernelGGL(victim_kernel, dim3(blocks), dim3(320), 0, sv, out, reps);
  CK(hipStreamSynchronize(sv));
  CK(hipMemcpy(ref.data(), out, (size_t)n * 4, hipMemcpyDeviceToHost));
  long launches = 0, bad_launches = 0, bad_words = 0;
  long lane_hist[64] = {0};
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    for (int i = 0; i < 6; i++) {
      if (kind == 0) hipLaunchKernelGGL((aggressor_kernel<0, 0>), dim3(1024), dim3(256), 0, sa, sink, 200);
      else if (kind == 1) hipLaunchKernelGGL((aggressor_kernel<1, 0>), dim3(1024), dim3(256), 0, sa, sink, 200);
      else if (kind == 2) hipLaunchKernelGGL((aggressor_kernel<2, 0>), dim3(1024), dim3(256), 0, sa, sink, 200);
      else if (kind == 3) hipLaunchKernelGGL((aggressor_kernel<3, 0>), dim3(1024), dim3(256), 0, sa, sink, 200);
      else if (kind == 10) hipLaunchKernelGGL((aggressor_kernel<0, 151>), dim3(1024), dim3(256), 0, sa, sink, 200);
      else if (kind == 11) hipLaunchKernelGGL((aggressor_kernel<0, 143>), dim3(1024), dim3(256), 0, sa, sink, 200);
      else if (kind == 12) hipLaunchKernelGGL((aggressor_kernel<0, 135>), dim3(1024), dim3(256), 0, sa, sink, 200);
      else if (kind == 13) hipLaunchKernelGGL((aggressor_kernel<3, 151>), dim3(1024), dim3(256), 0, sa, sink, 200);
      else if (kind >= 30 && kind <= 32) hipLaunchKernelGGL((aggressor_kernel<2, 151>), dim3(1024), dim3(256), 0, sa, sink, 200);
      else if (kind == 14) hipLaunchKernelGGL((aggressor_kernel<1, 151>), dim3(1024), dim3(256), 0, sa, sink, 200);
      else if (kind == 15) hipLaunchKernelGGL((aggressor_kernel<2, 151>), dim3(1024), dim3(256), 0, sa, sink, 200);
      else if (kind == 16) hipLaunchKernelGGL((aggressor_kernel<0, 127>), dim3(1024), dim3(256), 0, sa, sink, 200);
      else if (kind == 17) hipLaunchKernelGGL((aggressor_kernel<0, 119>), dim3(1024), dim3(256), 0, sa, sink, 200);
      else if (kind == 18) hipLaunchKernelGGL((aggressor_kernel<0, 95>), dim3(1024), dim3(256), 0, sa, sink, 200);
    }
    if (kind == 20) hipLaunchKernelGGL((mixed_kernel<0>), dim3(2 * blocks), dim3(320), 0, sv, out, reps, sink, 60);
    else if (kind == 21) hipLaunchKernelGGL((mixed_kernel<151>), dim3(2 * blocks), dim3(320), 0, sv, out, reps, sink, 60);
    else if (kind == 40) {
      // ONE stream: f16-MFMA kernel, packed-FP32 kernel, f16-MFMA kernel, back to back (what a single-threaded caller of the
      // library does all the time) -- in-order execution must keep them apart
      hipLaunchKernelGGL((aggressor_kernel<2, 151>), dim3(1024), dim3(256), 0, sv, sink, 50);
      hipLaunchKernelGGL((victim_kernel_t<151>), dim3(blocks), dim3(320), 0, sv, out, reps);
      hipLaunchKernelGGL((aggressor_kernel<2, 151>), dim3(1024), dim3(256), 0, sv, sink, 50);
    }
    else if (kind == 30) hipLaunchKernelGGL((victim_kernel_t<151>), dim3(blocks), dim3(320), 0, sv, out, reps);
    else if (kind == 31) hipLaunchKernelGGL((victim_kernel_t<127>), dim3(blocks), dim3(320), 0, sv, out, reps);
    else if (kind == 32) hipLaunchKernelGGL((victim_kernel_t<95>), dim3(blocks), dim3(320), 0, sv, out, reps);
    else hipLaunchKernelGGL(victim_kernel, dim3(blocks), dim3(320), 0, sv, out, reps);
    CK(hipStreamSynchronize(sv));
    CK(hipMemcpy(got.data(), out, (size_t)n * 4, hipMemcpyDeviceToHost));
    launches++;
    long b = 0;
    for (int i = 0; i < n; i++)
      if (memcmp(&got[i], &ref[i], 4) != 0) { b++; lane_hist[(i % 320) & 63]++; }
    if (b) { bad_launches++; bad_words += b; }
    CK(hipStreamSynchronize(sa));
  }
  printf("aggressor %d: %ld victim launches, %ld with wrong results (%ld wrong threads)\n", kind, launches, bad_launches, bad_words);
  if (bad_words) {
    printf("  wrong threads by lane of their wave:");
    for (int l = 0; l < 64; l++) printf(" %ld", lane_hist[l]);
    printf("\n");
  }
  return 0;
}
