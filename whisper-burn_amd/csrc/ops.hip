// K3/K6: LayerNorm and token embedding kernels (HBM-bound elementwise / row reductions).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "wave_ops.h"

namespace wb {
namespace {

// One wave per row.  Burn 0.9 nn::LayerNorm (used at /root/reference/src/model/mod.rs:155, :259,
// :300-301, :346-348): biased variance over the last dim, (x - mu) / (sqrt(var) + eps) * g + b;
// eps_inside_sqrt selects the later-Burn / PyTorch form (x - mu) / sqrt(var + eps).
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, float* __restrict__ y, int M,
                                                        int d, const float* __restrict__ g,
                                                        const float* __restrict__ b, float eps,
                                                        int eps_inside_sqrt) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * d);
  const int n4 = d >> 2;
  float s = 0.f;
  for (int c = lane; c < n4; c += 64) {
    float4 v = xr[c];
    s += (v.x + v.y) + (v.z + v.w);
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
  for (int c = lane; c < n4; c += 64) {
    float4 v = xr[c];
    float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
    q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
  }
  const float var = wave_sum(q) / (float)d;
  const float denom = eps_inside_sqrt ? sqrtf(var + eps) : (sqrtf(var) + eps);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  float4* yr = reinterpret_cast<float4*>(y + (int64_t)row * d);
  for (int c = lane; c < n4; c += 64) {
    float4 v = xr[c], gg = g4[c], bb = b4[c], o;
    o.x = (v.x - mean) / denom * gg.x + bb.x;
    o.y = (v.y - mean) / denom * gg.y + bb.y;
    o.z = (v.z - mean) / denom * gg.z + bb.z;
    o.w = (v.w - mean) / denom * gg.w + bb.w;
    yr[c] = o;
  }
}

// The same LayerNorm with the result written as fp16 pieces (kernels.h: GemmArgs::Ah): hi = fp16(y), lo = fp16((y - hi) 2^11)
__global__ __launch_bounds__(256) void layernorm_pieces_kernel(const float* __restrict__ x, unsigned short* __restrict__ yh,
                                                               unsigned short* __restrict__ yl, int M, int d,
                                                               const float* __restrict__ g, const float* __restrict__ b,
                                                               float eps, int eps_inside_sqrt) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * d);
  const int n4 = d >> 2;
  float s = 0.f;
  for (int c = lane; c < n4; c += 64) {
    float4 v = xr[c];
    s += (v.x + v.y) + (v.z + v.w);
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
  for (int c = lane; c < n4; c += 64) {
    float4 v = xr[c];
    float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
    q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
  }
  const float var = wave_sum(q) / (float)d;
  const float denom = eps_inside_sqrt ? sqrtf(var + eps) : (sqrtf(var) + eps);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  uint2* hr = reinterpret_cast<uint2*>(yh + (int64_t)row * d);
  uint2* lr = reinterpret_cast<uint2*>(yl + (int64_t)row * d);
  for (int c = lane; c < n4; c += 64) {
    float4 v = xr[c], gg = g4[c], bb = b4[c];
    const float o[4] = {(v.x - mean) / denom * gg.x + bb.x, (v.y - mean) / denom * gg.y + bb.y,
                        (v.z - mean) / denom * gg.z + bb.z, (v.w - mean) / denom * gg.w + bb.w};
    unsigned hb[4], lb[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const _Float16 hh = (_Float16)o[i];
      hb[i] = __builtin_bit_cast(unsigned short, hh);
      lb[i] = __builtin_bit_cast(unsigned short, (_Float16)((o[i] - (float)hh) * 2048.f));
    }
    hr[c] = make_uint2(hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16));
    lr[c] = make_uint2(lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16));
  }
}

// x[r] = E[tok[r]] + pos[r % L]   (mod.rs:141-146)
__global__ void embed_kernel(const int32_t* __restrict__ tok, int n_rows, int L, int d,
                             const float* __restrict__ E, const float* __restrict__ pos, float* __restrict__ x) {
  const int r = blockIdx.x;
  const float4* e = reinterpret_cast<const float4*>(E + (int64_t)tok[r] * d);
  const float4* p = reinterpret_cast<const float4*>(pos + (int64_t)(r % L) * d);
  float4* o = reinterpret_cast<float4*>(x + (int64_t)r * d);
  for (int c = threadIdx.x; c < (d >> 2); c += blockDim.x) {
    float4 a = e[c], b = p[c];
    o[c] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

}  // namespace

void launch_layernorm(hipStream_t st, const float* x, float* y, int M, int d, const float* g, const float* b,
                      float eps, int eps_inside_sqrt) {
  if (M <= 0) return;
  hipLaunchKernelGGL(layernorm_kernel, dim3((M + 3) / 4), dim3(256), 0, st, x, y, M, d, g, b, eps, eps_inside_sqrt);
}

void launch_layernorm_pieces(hipStream_t st, const float* x, uint16_t* yh, uint16_t* yl, int M, int d, const float* g,
                             const float* b, float eps, int eps_inside_sqrt) {
  if (M <= 0) return;
  hipLaunchKernelGGL(layernorm_pieces_kernel, dim3((M + 3) / 4), dim3(256), 0, st, x, yh, yl, M, d, g, b, eps, eps_inside_sqrt);
}

void launch_embed(hipStream_t st, const int32_t* tok, int n_rows, int L, int d, const float* E, const float* pos,
                  float* x) {
  if (n_rows <= 0) return;
  hipLaunchKernelGGL(embed_kernel, dim3(n_rows), dim3(128), 0, st, tok, n_rows, L, d, E, pos, x);
}

}  // namespace wb
