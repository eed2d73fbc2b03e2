// Developer probe: encoder self-attention at the bench geometry (3 windows of tiny.en), 128-query blocks vs the
// key-split variant; prints time and the largest difference between the two.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip tools/attn_probe.cpp -o tools/attn_probe
#include "../csrc/attention.hip"
#include <cstdio>
#include <vector>
using namespace wb;
static void run(const char* name, const std::vector<int>& lens, int H) {
  const int d = 64 * H, nw = (int)lens.size();
  int rows = 0, maxq = 0;
  std::vector<AttnSeg> segs;
  for (int c : lens) { segs.push_back(AttnSeg{rows, c, rows, c}); rows += c; maxq = std::max(maxq, c); }
  std::vector<float> h((size_t)rows * 3 * d);
  for (size_t i = 0; i < h.size(); i++) h[i] = (float)((int)((i * 2654435761u) >> 18 & 1023) - 512) * (1.0f / 512.0f);
  float *qkv, *o1, *o2; AttnSeg* ds;
  hipMalloc(&qkv, h.size() * 4); hipMalloc(&o1, (size_t)rows * d * 4); hipMalloc(&o2, (size_t)rows * d * 4);
  hipMalloc(&ds, segs.size() * sizeof(AttnSeg));
  hipMemcpy(qkv, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(ds, segs.data(), segs.size() * sizeof(AttnSeg), hipMemcpyHostToDevice);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const float scale = 0.35355339f;
  auto a = [&](float* o) {
    dim3 grid((maxq + 127) / 128, H, nw);
    hipLaunchKernelGGL((attention_f32_kernel<4>), grid, dim3(256), 0, st, qkv, 3 * d, qkv + d, qkv + 2 * d, 3 * d, o, d, ds, scale, 0);
  };
  auto b = [&](float* o) {
    dim3 grid((maxq + 31) / 32, H, nw);
    hipLaunchKernelGGL((attention_f32_kvsplit_kernel<4>), grid, dim3(256), 0, st, qkv, 3 * d, qkv + d, qkv + 2 * d, 3 * d, o, d, ds, scale, 0);
  };
  hipMemsetAsync(o1, 0, (size_t)rows * d * 4, st); hipMemsetAsync(o2, 0, (size_t)rows * d * 4, st);
  a(o1); b(o2);
  hipStreamSynchronize(st);
  printf("%s: launch status %s\n", name, hipGetErrorString(hipGetLastError()));
  std::vector<float> r1((size_t)rows * d), r2((size_t)rows * d);
  hipMemcpy(r1.data(), o1, r1.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), o2, r2.size() * 4, hipMemcpyDeviceToHost);
  double md = 0, mx = 0;
  for (size_t i = 0; i < r1.size(); i++) { md = std::max(md, (double)fabsf(r1[i] - r2[i])); mx = std::max(mx, (double)fabsf(r1[i])); }
  float ms1, ms2;
  for (int i = 0; i < 3; i++) { a(o1); b(o2); }
  hipEventRecord(e0, st); for (int i = 0; i < 20; i++) a(o1); hipEventRecord(e1, st); hipStreamSynchronize(st); hipEventElapsedTime(&ms1, e0, e1);
  hipEventRecord(e0, st); for (int i = 0; i < 20; i++) b(o2); hipEventRecord(e1, st); hipStreamSynchronize(st); hipEventElapsedTime(&ms2, e0, e1);
  double flop = 0; for (int c : lens) flop += 4.0 * c * c * 64 * H;
  printf("%s: rows %d heads %d: blocks128 %.1f us (%.1f TF/s), key-split %.1f us (%.1f TF/s), max |diff| %.3g of max |o| %.3g\n", name, rows, H,
         ms1 * 50, flop / (ms1 * 50e-6) / 1e12, ms2 * 50, flop / (ms2 * 50e-6) / 1e12, md, mx);
  hipFree(qkv); hipFree(o1); hipFree(o2); hipFree(ds);
}
int main() {
  run("tiny.en 30 s", {745, 745, 324}, 6);
  run("base.en 30 s", {745, 745, 324}, 8);
  run("one ragged window", {387}, 6);
  run("large-v2 window", {745}, 20);
  run("small 8 windows", {745, 745, 745, 745, 745, 745, 745, 745}, 12);
  return 0;
}
