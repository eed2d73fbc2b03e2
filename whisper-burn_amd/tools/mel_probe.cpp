// Developer probe: mel kernel + clamp fix-up time on a batch of reference windows.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip tools/mel_probe.cpp -o tools/mel_probe
// (To time the kernel up to stage N, cut a private copy of csrc/mel.hip: the shipped kernel carries no probe switches.)
#include "../csrc/mel.hip"
#include <cstdio>
#include <vector>
using namespace wb;
int main() {
  const int W = 256, N = 238559, T = N / 160;
  float* pcm; hipMalloc(&pcm, (size_t)W * N * 4);
  std::vector<float> h((size_t)N);
  for (int i = 0; i < N; i++) h[i] = 0.1f * sinf(0.01f * i) + 0.001f * (i % 97);
  for (int w = 0; w < W; w++) hipMemcpy(pcm + (size_t)w * N, h.data(), (size_t)N * 4, hipMemcpyHostToDevice);
  std::vector<MelWindow> wins(W);
  for (int w = 0; w < W; w++) wins[w] = MelWindow{(int64_t)w * N, N, T, T, 0};
  MelWindow* dw; hipMalloc(&dw, W * sizeof(MelWindow)); hipMemcpy(dw, wins.data(), W * sizeof(MelWindow), hipMemcpyHostToDevice);
  MelTables ht; mel_tables_build(16000.0, &ht);
  MelTables* dt; hipMalloc(&dt, sizeof(MelTables)); hipMemcpy(dt, &ht, sizeof(MelTables), hipMemcpyHostToDevice);
  const int Ts = (T + 3) & ~3;
  float* out; hipMalloc(&out, (size_t)W * 80 * Ts * 4);
  float* gmax; hipMalloc(&gmax, (size_t)W * mel_bmax_stride(T) * 2 * 4);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < 3; it++) {
    hipEventRecord(a, st);
    launch_mel_spectrogram(st, pcm, dw, W, T, dt, out, (int64_t)80 * Ts, Ts, gmax, 0, Ts);
    launch_mel_finalize(st, dw, W, out, (int64_t)80 * Ts, Ts, gmax, T);
    hipEventRecord(b, st);
    hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (it == 2) printf("%d windows, %d frames: %.1f us -> %.3f G frames/s, %.1f GB/s algorithmic\n", W, W * T, ms * 1e3, W * T / (ms * 1e-3) / 1e9, 960.0 * W * T / (ms * 1e-3) / 1e9);
  }
  return 0;
}
