// Developer probe: resident blocks per CU of the mel kernel as the runtime sees them (LDS allocation granularity).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip tools/occ_probe.cpp -o tools/occ_probe
#include "../csrc/mel.hip"
#include <cstdio>
using namespace wb;
__global__ void lds_k(float* o, int n) { extern __shared__ float sm[]; sm[threadIdx.x] = 1.f; __syncthreads(); if (n == 12345) o[0] = sm[0]; }
int main() {
  int nb = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, mel_spectrogram_kernel, MEL_THREADS, 0);
  printf("mel_spectrogram_kernel: %d blocks/CU by the occupancy API (static LDS %d B)\n", nb, (int)(PAIRS * 2 * FROW * 4));
  for (int bytes : {54272, 54400, 54528, 54784, 55296, 65536, 81920}) {
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lds_k, 320, bytes);
    printf("  320 threads, %6d B dynamic LDS -> %d blocks/CU\n", bytes, nb);
  }
  return 0;
}
