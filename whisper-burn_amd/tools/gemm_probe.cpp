// Developer probe: the exact-f32 MFMA GEMM at the encoder's shapes, one line per (shape, tile/k-depth/prefetch config).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWB_GEMM_PROBE -x hip tools/gemm_probe.cpp -o tools/gemm_probe
// Every configuration keeps the k order of the accumulation chain, so outputs must be bit-identical: checked.
#include "../csrc/gemm.hip"
#include <cstdio>
#include <cstring>
#include <vector>
using namespace wb;
int main() {
  struct Shape { const char* name; int M, K, N; };
  const Shape shapes[] = {
      {"tiny qkv", 1814, 384, 1152}, {"tiny out", 1814, 384, 384},   {"tiny mlp1", 1814, 384, 1536},
      {"tiny mlp2", 1814, 1536, 384}, {"tiny conv2", 1814, 1152, 384}, {"base mlp1", 1814, 512, 2048},
      {"small qkv", 11920, 768, 2304}, {"small out", 11920, 768, 768}, {"small mlp2", 11920, 3072, 768},
      {"large qkv", 11920, 1280, 3840}, {"large mlp2", 11920, 5120, 1280}};
  const int NCFG = 14, REP = 10;
  size_t maxA = 0, maxB = 0, maxC = 0;
  for (auto& s : shapes) { maxA = std::max(maxA, (size_t)s.M * s.K); maxB = std::max(maxB, (size_t)s.K * s.N); maxC = std::max(maxC, (size_t)s.M * s.N); }
  float *A, *B, *C, *C0, *bias;
  hipMalloc(&A, maxA * 4); hipMalloc(&B, maxB * 4); hipMalloc(&C, maxC * 4); hipMalloc(&C0, maxC * 4); hipMalloc(&bias, 8192 * 4);
  std::vector<float> h(std::max(maxA, maxB));
  for (size_t i = 0; i < h.size(); i++) h[i] = (float)((int)((i * 2654435761u) >> 20 & 255) - 128) * 0.01f;
  hipMemcpy(A, h.data(), maxA * 4, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), maxB * 4, hipMemcpyHostToDevice);
  hipMemcpy(bias, h.data(), 8192 * 4, hipMemcpyHostToDevice);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ref(maxC), got(maxC);
  for (auto& s : shapes) {
    GemmArgs g;
    g.A = A; g.lda = s.K; g.B = B; g.ldb = s.N; g.C = C; g.ldc = s.N; g.bias = bias; g.M = s.M; g.N = s.N; g.K = s.K; g.act = ACT_GELU;
    const double flop = 2.0 * s.M * s.K * s.N;
    for (int cfg = 0; cfg < NCFG; cfg++) {
      hipMemsetAsync(C, 0, (size_t)s.M * s.N * 4, st);
      if (launch_gemm_f32_cfg(st, g, cfg) != 0) continue;
      if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { printf("%s cfg %d: launch failed\n", s.name, cfg); continue; }
      hipMemcpy(got.data(), C, (size_t)s.M * s.N * 4, hipMemcpyDeviceToHost);
      if (cfg == 0) ref = got;
      const bool same = memcmp(ref.data(), got.data(), (size_t)s.M * s.N * 4) == 0;
      for (int i = 0; i < 3; i++) launch_gemm_f32_cfg(st, g, cfg);
      hipEventRecord(e0, st);
      for (int i = 0; i < REP; i++) launch_gemm_f32_cfg(st, g, cfg);
      hipEventRecord(e1, st);
      hipStreamSynchronize(st);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%-11s M=%5d K=%4d N=%4d cfg %2d: %8.1f us  %6.1f TF/s  %s\n", s.name, s.M, s.K, s.N, cfg, ms * 1e3 / REP,
             flop / (ms * 1e-3 / REP) / 1e12, same ? "bit-identical" : "DIFFERS");
    }
  }
  return 0;
}
