// wave64 cross-lane helpers on DPP / v_permlane32_swap instead of ds_bpermute.
//
// hipcc lowers every __shfl_xor to ds_bpermute_b32 (an LDS-crossbar op, ~100+ cycles of dependent
// latency each); a 6-step butterfly reduction is then ~1 us of pure latency, which is as much as
// the rest of a latency-bound decode kernel.  DPP modifiers ride on the VALU op itself.
#pragma once
#include <hip/hip_runtime.h>

namespace wb {

// DPP controls (gfx9 encoding)
constexpr int DPP_QUAD_XOR1 = 0xB1;        // quad_perm [1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;        // quad_perm [2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_BCAST15 = 0x142;
constexpr int DPP_ROW_BCAST31 = 0x143;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float identity, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xF,
                                                    false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int identity, int v) {
  return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xF, false);
}

__device__ __forceinline__ float readlane63(float v) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Sum over the 64 lanes, returned in every lane (all lanes must be active).
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<DPP_QUAD_XOR1, 0xF>(0.f, v);
  v += dpp_f<DPP_QUAD_XOR2, 0xF>(0.f, v);
  v += dpp_f<DPP_ROW_HALF_MIRROR, 0xF>(0.f, v);
  v += dpp_f<DPP_ROW_MIRROR, 0xF>(0.f, v);
  v += dpp_f<DPP_ROW_BCAST15, 0xA>(0.f, v);
  v += dpp_f<DPP_ROW_BCAST31, 0xC>(0.f, v);
  return readlane63(v);
}

__device__ __forceinline__ float wave_max(float v) {
  const float ninf = -INFINITY;
  v = fmaxf(v, dpp_f<DPP_QUAD_XOR1, 0xF>(ninf, v));
  v = fmaxf(v, dpp_f<DPP_QUAD_XOR2, 0xF>(ninf, v));
  v = fmaxf(v, dpp_f<DPP_ROW_HALF_MIRROR, 0xF>(ninf, v));
  v = fmaxf(v, dpp_f<DPP_ROW_MIRROR, 0xF>(ninf, v));
  v = fmaxf(v, dpp_f<DPP_ROW_BCAST15, 0xA>(ninf, v));
  v = fmaxf(v, dpp_f<DPP_ROW_BCAST31, 0xC>(ninf, v));
  return readlane63(v);
}

// (value desc, id asc) order used by every top-k in the engine
__device__ __forceinline__ bool better(float v, int id, float bv, int bid) {
  return v > bv || (v == bv && id < bid);
}

// Best (value, id) pair over the 64 lanes, returned in every lane.
__device__ __forceinline__ void wave_argmax(float& v, int& id) {
  const float ninf = -INFINITY;
  const int imax = 0x7fffffff;
#define WB_STEP(CTRL, MASK)                                                 \
  {                                                                         \
    const float ov = dpp_f<CTRL, MASK>(ninf, v);                            \
    const int oi = dpp_i<CTRL, MASK>(imax, id);                             \
    if (better(ov, oi, v, id)) { v = ov; id = oi; }                         \
  }
  WB_STEP(DPP_QUAD_XOR1, 0xF)
  WB_STEP(DPP_QUAD_XOR2, 0xF)
  WB_STEP(DPP_ROW_HALF_MIRROR, 0xF)
  WB_STEP(DPP_ROW_MIRROR, 0xF)
  WB_STEP(DPP_ROW_BCAST15, 0xA)
  WB_STEP(DPP_ROW_BCAST31, 0xC)
#undef WB_STEP
  v = readlane63(v);
  id = __builtin_amdgcn_readlane(id, 63);
}

// v[lane] + v[lane ^ 32] in every lane (v_permlane32_swap: the two half-waves trade places).
__device__ __forceinline__ float xor32_sum(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// v[lane] + v[lane ^ 16] in every lane (v_permlane16_swap: the rows of each row pair trade places).
__device__ __forceinline__ float xor16_sum(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_max(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// sum / max over the 16 lanes of a DPP row (lanes 16r .. 16r+15), returned in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<DPP_QUAD_XOR1, 0xF>(0.f, v);
  v += dpp_f<DPP_QUAD_XOR2, 0xF>(0.f, v);
  v += dpp_f<DPP_ROW_HALF_MIRROR, 0xF>(0.f, v);
  v += dpp_f<DPP_ROW_MIRROR, 0xF>(0.f, v);
  return v;
}
__device__ __forceinline__ float xor32_max(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

}  // namespace wb
