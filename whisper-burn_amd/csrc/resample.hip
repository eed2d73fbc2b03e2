// Polyphase sample-rate conversion in HBM (next to the hot path: SURVEY 8(f3)).
//
// The reference has no resampler: its CLI rejects anything but 16 kHz (bin/transcribe/main.rs:42) and the README
// sends other files through `sox` (README.md:69-74) -- including the bundled 22 050 Hz audio.wav.  This is the
// standard rational resampler (upsample by `up`, Kaiser-windowed sinc low-pass, keep every `down`-th sample) with
// the design every SciPy user gets from `resample_poly(x, up, down)`: cut-off 1/max(up,down) of Nyquist, half
// length 10*max(up,down), Kaiser beta 5, unit DC gain times `up`, output j centred on input time j*down/up, zeros
// outside the signal.  The taps are designed in f64 on the host, the convolution runs in f32:
//   y[j] = sum_i x[i] * h[j*down + half - i*up],   0 <= j*down + half - i*up <= 2*half,   n_out = ceil(n_in*up/down)
// One pass, HBM-bound: 4 B read per input sample + 4 B written per output sample; the taps sit in LDS.
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>

#include "kernels.h"
#include "wb_internal.h"

namespace wb {
namespace {

constexpr int RS_MAX_RATE = 1600;            // taps = 20*max(up,down)+1 floats must fit the 160 KB LDS (<= 128 KB here)
constexpr int RS_THREADS = 256, RS_PER_THREAD = 8;

double bessel_i0(double x) {                 // power series; x <= 5 here, converges in ~20 terms
  const double q = 0.25 * x * x;
  double term = 1.0, sum = 1.0;
  for (int k = 1; k < 64; k++) {
    term *= q / ((double)k * (double)k);
    sum += term;
    if (term < 1e-17 * sum) break;
  }
  return sum;
}

// firwin(2*half+1, 1/max_rate, window=("kaiser", 5.0)) * up
void design_taps(int up, int down, std::vector<float>* taps, int* half_out) {
  const int max_rate = up > down ? up : down;
  const int half = 10 * max_rate, n = 2 * half + 1;
  const double fc = 1.0 / (double)max_rate, beta = 5.0, i0b = bessel_i0(beta), pi = 3.14159265358979323846;
  std::vector<double> h((size_t)n);
  double sum = 0.0;
  for (int k = 0; k < n; k++) {
    const double m = (double)(k - half), s = pi * fc * m;
    const double sinc = (m == 0.0) ? 1.0 : std::sin(s) / s;
    const double r = m / (double)half;
    const double w = bessel_i0(beta * std::sqrt(std::fmax(0.0, 1.0 - r * r))) / i0b;
    h[(size_t)k] = fc * sinc * w;
    sum += h[(size_t)k];
  }
  taps->resize((size_t)n);
  for (int k = 0; k < n; k++) (*taps)[(size_t)k] = (float)(h[(size_t)k] / sum * (double)up);
  *half_out = half;
}

int reduce_rates(int32_t rate_in, int32_t rate_out, int* up, int* down) {
  WB_REQUIRE(rate_in > 0 && rate_out > 0, WB_ERR_ARG, "resample: sample rates must be positive (%d -> %d)", rate_in, rate_out);
  const int g = std::gcd(rate_in, rate_out);
  *up = rate_out / g;
  *down = rate_in / g;
  WB_REQUIRE(*up <= RS_MAX_RATE && *down <= RS_MAX_RATE, WB_ERR_SHAPE,
             "resample: %d -> %d Hz reduces to %d/%d; ratios above %d are not supported", rate_in, rate_out, *up, *down,
             RS_MAX_RATE);
  return WB_OK;
}

__global__ __launch_bounds__(RS_THREADS) void resample_poly_kernel(const float* __restrict__ x, int64_t n_in,
                                                                    const float* __restrict__ taps, int half, int up,
                                                                    int down, float* __restrict__ y, int64_t n_out) {
  HIP_DYNAMIC_SHARED(float, h)
  const int n_taps = 2 * half + 1;
  for (int k = threadIdx.x; k < n_taps; k += RS_THREADS) h[k] = taps[k];
  __syncthreads();
  const int64_t j0 = (int64_t)blockIdx.x * (RS_THREADS * RS_PER_THREAD);
#pragma unroll 1
  for (int r = 0; r < RS_PER_THREAD; r++) {
    const int64_t j = j0 + r * RS_THREADS + threadIdx.x;       // consecutive lanes -> consecutive outputs
    if (j >= n_out) break;
    const int64_t t = j * down + half;                         // position on the up-sampled grid
    int64_t i = t / up;                                        // newest input sample under the filter
    int k = (int)(t - i * up);                                 // its tap
    if (i >= n_in) {                                           // zeros beyond the end of the signal
      const int64_t skip = i - (n_in - 1);
      i -= skip;
      k += (int)skip * up;
    }
    float acc = 0.f;
    for (; k < n_taps && i >= 0; k += up, i--) acc = fmaf(x[i], h[k], acc);
    y[j] = acc;
  }
}

struct TapCache {
  std::mutex mu;
  std::map<std::tuple<int, int, int>, std::pair<float*, int>> dev;      // (device, up, down) -> (taps, half)
};
TapCache& tap_cache() { static TapCache c; return c; }

int device_taps(int device, int up, int down, const float** taps, int* half) {
  TapCache& c = tap_cache();
  std::lock_guard<std::mutex> lk(c.mu);
  auto key = std::make_tuple(device, up, down);
  auto it = c.dev.find(key);
  if (it == c.dev.end()) {
    std::vector<float> host;
    int hl = 0;
    design_taps(up, down, &host, &hl);
    float* d = nullptr;
    WB_HIP(hipMalloc(&d, host.size() * sizeof(float)));
    WB_HIP(hipMemcpy(d, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    it = c.dev.emplace(key, std::make_pair(d, hl)).first;
  }
  *taps = it->second.first;
  *half = it->second.second;
  return WB_OK;
}

}  // namespace
}  // namespace wb

using namespace wb;

extern "C" {

int64_t wb_resample_len(int64_t n_in, int32_t rate_in, int32_t rate_out) {
  int up, down;
  if (n_in < 0 || reduce_rates(rate_in, rate_out, &up, &down) != WB_OK) return -1;
  return (n_in * up + down - 1) / down;
}

int wb_resample_filter(int32_t rate_in, int32_t rate_out, float* taps, int32_t capacity, int32_t* n_taps, int32_t* up_out,
                       int32_t* down_out) {
  int up, down, half;
  WB_TRY(reduce_rates(rate_in, rate_out, &up, &down));
  std::vector<float> h;
  design_taps(up, down, &h, &half);
  if (n_taps) *n_taps = (int32_t)h.size();
  if (up_out) *up_out = up;
  if (down_out) *down_out = down;
  if (taps) {
    WB_REQUIRE((int64_t)h.size() <= capacity, WB_ERR_ARG, "wb_resample_filter: %zu taps > capacity %d", h.size(), capacity);
    memcpy(taps, h.data(), h.size() * sizeof(float));
  }
  return WB_OK;
}

int wb_resample_dev(int device, const float* src_dev, int64_t n_in, int32_t rate_in, int32_t rate_out, float* dst_dev,
                    int64_t capacity, int64_t* n_out) {
  int up, down;
  WB_TRY(reduce_rates(rate_in, rate_out, &up, &down));
  WB_REQUIRE(n_in >= 0 && (n_in == 0 || (src_dev && dst_dev)), WB_ERR_ARG, "wb_resample_dev: bad argument");
  const int64_t n = (n_in * up + down - 1) / down;
  WB_REQUIRE(n <= capacity, WB_ERR_ARG, "wb_resample_dev: %lld output samples > capacity %lld", (long long)n,
             (long long)capacity);
  if (n_out) *n_out = n;
  if (n == 0) return WB_OK;
  wb::GpuTurn turn(device);
  WB_HIP(hipSetDevice(device));
  if (up == 1 && down == 1) {
    WB_HIP(hipMemcpyAsync(dst_dev, src_dev, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, nullptr));
  } else {
    const float* taps = nullptr;
    int half = 0;
    WB_TRY(device_taps(device, up, down, &taps, &half));
    const size_t lds = (size_t)(2 * half + 1) * sizeof(float);
    if (lds > 48 * 1024)
      WB_HIP(hipFuncSetAttribute((const void*)resample_poly_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t per_block = RS_THREADS * RS_PER_THREAD;
    hipLaunchKernelGGL(resample_poly_kernel, dim3((unsigned)((n + per_block - 1) / per_block)), dim3(RS_THREADS), lds,
                       nullptr, src_dev, n_in, taps, half, up, down, dst_dev, n);
    WB_HIP(hipGetLastError());
  }
  WB_HIP(hipStreamSynchronize(nullptr));
  return WB_OK;
}

}  // extern "C"
