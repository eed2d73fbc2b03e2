// hipemu -- a FUNCTIONAL model of the gfx950 execution environment for kernel development on a machine
// without a GPU.  TEST INFRASTRUCTURE, not a product path: it is only ever compiled into
// lib/libwhisper_hip_emu.so by tools/hipemu/Makefile, the Python binding refuses that library unless a test
// sets WHISPER_HIP_ALLOW_EMU=1, and bench.py / smoke() / every `-m gpu` test require a real cuda:0.
//
// What it models: a launch runs block after block; the threads of a block are ucontext fibers that run until
// they reach a block barrier (__syncthreads) or a wave-collective operation (DPP, v_readlane, v_permlane*_swap,
// MFMA, ...), where the 64 lanes of the wave exchange operands exactly as the instruction does (lane / row /
// bank semantics of the ISA; the MFMA register layouts of the CDNA matrix cores).  __shared__ variables are
// thread-local statics: an ordinary launch runs all its blocks on the calling thread, one at a time; a CO-RESIDENT
// launch (hipemu::launch_coop, the model of a cooperative / persistent grid whose blocks wait for one another on
// device-scope counters) gives every block its own OS thread -- hence its own LDS -- and passes a baton between
// them, so exactly one block runs at any time and a block that spins (s_sleep) hands the baton on.
// Device memory is host memory; streams are synchronous; stream capture records closures that hipGraphLaunch replays.
//
// What it does NOT model: timing, caches, memory ordering between blocks, LDS capacity, register pressure.
// Code that relies on the lock-step of a wave WITHOUT a collective or a barrier (wave-local LDS exchange) must
// mark the exchange point with __builtin_amdgcn_wave_barrier() (a scheduling no-op on the GPU).
#pragma once
#define HIPEMU 1
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <functional>
#include <tuple>
#include <type_traits>
#include <utility>

// ---- language qualifiers ------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __align__(n) __attribute__((aligned(n)))
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(hipemu::dyn_shared());
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
template <class T, class U> static inline void hipemu_atomic_store(T* p, U v) { T t = (T)v; __atomic_store(p, &t, __ATOMIC_SEQ_CST); }
template <class T> static inline T hipemu_atomic_load(const T* p) { T v; __atomic_load(const_cast<T*>(p), &v, __ATOMIC_SEQ_CST); return v; }
#define __hip_atomic_store(p, v, order, scope) hipemu_atomic_store((p), (v))
#define __hip_atomic_load(p, order, scope) hipemu_atomic_load((p))
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), (order))
// hooks of the engine's cross-block hand-off helpers (csrc/wave_ops.h defines the gfx950 forms unless these exist)
#define WB_DRAIN_VMEM() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define WB_LAUNDER_V(x) ((void)0)
#define WB_PIN_F4(v) ((void)0)
#define WB_LAUNCH_COOP(kernel, grid, block, shmem, stream, arg) \
  (hipemu::launch_coop(kernel, dim3(grid), dim3(block), (size_t)(shmem), stream, arg), hipSuccess)

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---- vector types -------------------------------------------------------------------------------------------
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

// ---- runtime types ------------------------------------------------------------------------------------------
typedef int hipError_t;
enum : int {
  hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorUnknown = 999
};
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2,
                     hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
constexpr unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocMapped = 2, hipHostMallocDefault = 0;
struct ihipStream_t; struct ihipEvent_t; struct ihipGraph; struct ihipGraphExec; struct ihipGraphNode;
typedef ihipStream_t* hipStream_t;
typedef ihipEvent_t* hipEvent_t;
typedef ihipGraph* hipGraph_t;
typedef ihipGraphExec* hipGraphExec_t;
typedef ihipGraphNode* hipGraphNode_t;

hipError_t hipSetDevice(int);
hipError_t hipGetDevice(int*);
hipError_t hipGetDeviceCount(int*);
hipError_t hipDeviceSynchronize();
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t);
hipError_t hipMalloc(void**, size_t);
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
hipError_t hipFree(void*);
hipError_t hipHostMalloc(void**, size_t, unsigned);
hipError_t hipHostFree(void*);
hipError_t hipHostGetDevicePointer(void**, void*, unsigned);
hipError_t hipMemcpy(void*, const void*, size_t, hipMemcpyKind);
hipError_t hipMemcpyAsync(void*, const void*, size_t, hipMemcpyKind, hipStream_t);
hipError_t hipMemcpy2DAsync(void*, size_t, const void*, size_t, size_t, size_t, hipMemcpyKind, hipStream_t);
hipError_t hipMemset(void*, int, size_t);
hipError_t hipMemsetAsync(void*, int, size_t, hipStream_t);
hipError_t hipStreamCreate(hipStream_t*);
hipError_t hipStreamCreateWithFlags(hipStream_t*, unsigned);
hipError_t hipStreamDestroy(hipStream_t);
hipError_t hipStreamSynchronize(hipStream_t);
hipError_t hipStreamQuery(hipStream_t);
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned);
hipError_t hipEventCreate(hipEvent_t*);
hipError_t hipEventCreateWithFlags(hipEvent_t*, unsigned);
hipError_t hipEventDestroy(hipEvent_t);
hipError_t hipEventRecord(hipEvent_t, hipStream_t);
hipError_t hipEventSynchronize(hipEvent_t);
hipError_t hipEventElapsedTime(float*, hipEvent_t, hipEvent_t);
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode);
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*);
hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, hipGraphNode_t*, char*, size_t);
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t);
hipError_t hipGraphDestroy(hipGraph_t);
hipError_t hipGraphExecDestroy(hipGraphExec_t);
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int);
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 16, hipDeviceAttributeCooperativeLaunch = 95 };
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t attr, int device);   // CU count: HIPEMU_CUS (default 256)
// resident blocks per CU of a kernel: the model has no registers or LDS budget -- HIPEMU_BLOCKS_PER_CU (default 1)
hipError_t hipemu_occupancy(int* blocks_per_cu);
#define hipOccupancyMaxActiveBlocksPerMultiprocessor(out, kernel, block, shmem) hipemu_occupancy(out)

// ---- execution model ----------------------------------------------------------------------------------------
namespace hipemu {
struct Fiber;
struct ThreadCtx { dim3 tidx, bidx, bdim, gdim; };
extern thread_local ThreadCtx* g_tc;   // the running fiber's coordinates
void* dyn_shared();
void syncthreads();
enum Op { OP_WAVE_BARRIER = 1, OP_READFIRSTLANE, OP_READLANE, OP_DPP, OP_PERMLANE32_SWAP, OP_PERMLANE16_SWAP,
          OP_MFMA_F32_32X32X2, OP_MFMA_BF16_32X32X16, OP_MFMA_F32_16X16X4, OP_MFMA_F16_16X16X32, OP_SHFL, OP_SHFL_XOR, OP_BALLOT };
// hands the calling lane's operands to the wave and returns when the collective has been executed
void wave_op(int op, const void* in0, const void* in1, const void* in2, void* out, int i0, int i1, int i2, int i3);
void enqueue(hipStream_t st, std::function<void()> body, dim3 grid, dim3 block, size_t shmem, hipEvent_t e0, hipEvent_t e1,
             bool coresident = false);
// a spinning thread gives way: inside a co-resident launch the block's baton moves on once no thread of the block
// can run; in an ordinary launch (blocks run to completion one after another) a spin can never be satisfied -> abort
void spin_yield();

template <class... P, class... A>
void launch(void (*k)(P...), dim3 g, dim3 b, size_t sh, hipStream_t st, hipEvent_t e0, hipEvent_t e1, A&&... a) {
  std::tuple<std::decay_t<P>...> args(std::forward<A>(a)...);
  enqueue(st, [k, args]() { std::apply(k, args); }, g, b, sh, e0, e1);
}
template <class... P, class... A>
void launch_coop(void (*k)(P...), dim3 g, dim3 b, size_t sh, hipStream_t st, A&&... a) {
  std::tuple<std::decay_t<P>...> args(std::forward<A>(a)...);
  enqueue(st, [k, args]() { std::apply(k, args); }, g, b, sh, nullptr, nullptr, true);
}
}  // namespace hipemu

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hipemu::launch(kernel, dim3(grid), dim3(block), (size_t)(shmem), stream, nullptr, nullptr, ##__VA_ARGS__)
#define hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, ev0, ev1, flags, ...) \
  hipemu::launch(kernel, dim3(grid), dim3(block), (size_t)(shmem), stream, ev0, ev1, ##__VA_ARGS__)

#define threadIdx (hipemu::g_tc->tidx)
#define blockIdx (hipemu::g_tc->bidx)
#define blockDim (hipemu::g_tc->bdim)
#define gridDim (hipemu::g_tc->gdim)
constexpr int warpSize = 64;

static inline void __syncthreads() { hipemu::syncthreads(); }
static inline unsigned long long wall_clock64() { return 0; }   // (no timing in the model)
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

// ---- scalar intrinsics --------------------------------------------------------------------------------------
static inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
static inline float __uint_as_float(unsigned v) { float f; memcpy(&f, &v, 4); return f; }
static inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
static inline unsigned __float_as_uint(float f) { unsigned v; memcpy(&v, &f, 4); return v; }
static inline float __fdiv_rn(float a, float b) { return a / b; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.0f / a; }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
#define __expf expf
#define __logf logf
#define __log2f log2f
#define __exp2f exp2f
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline long min(long a, long b) { return a < b ? a : b; }
static inline long max(long a, long b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline unsigned long min(unsigned long a, unsigned long b) { return a < b ? a : b; }
static inline unsigned long max(unsigned long a, unsigned long b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline float rsqrtf(float a) { return 1.0f / sqrtf(a); }
static inline float __fsqrt_rn(float a) { return sqrtf(a); }
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> static inline T atomicCAS(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }

// ---- buffer resources (raw 16-byte loads / stores with cache-policy bits: plain memory here) ---------------------
struct hipemu_rsrc { char* base; };
typedef hipemu_rsrc __amdgpu_buffer_rsrc_t;
typedef unsigned hipemu_u32x4 __attribute__((ext_vector_type(4)));
static inline hipemu_rsrc hipemu_make_buffer_rsrc(const void* p, int, int, int) { return hipemu_rsrc{(char*)p}; }
static inline hipemu_u32x4 hipemu_raw_buffer_load_b128(hipemu_rsrc r, unsigned voff, unsigned soff, int) {
  hipemu_u32x4 v; memcpy(&v, r.base + voff + soff, 16); return v;
}
static inline void hipemu_raw_buffer_store_b128(hipemu_u32x4 v, hipemu_rsrc r, unsigned voff, unsigned soff, int) {
  memcpy(r.base + voff + soff, &v, 16);
}
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, n, flags) hipemu_make_buffer_rsrc((const void*)(p), stride, n, flags)
static inline unsigned hipemu_raw_buffer_load_b32(hipemu_rsrc r, unsigned voff, unsigned soff, int) {
  unsigned v; memcpy(&v, r.base + voff + soff, 4); return v;
}
typedef unsigned hipemu_u32x2v __attribute__((ext_vector_type(2)));
static inline hipemu_u32x2v hipemu_raw_buffer_load_b64(hipemu_rsrc r, unsigned voff, unsigned soff, int) {
  hipemu_u32x2v v; memcpy(&v, r.base + voff + soff, 8); return v;
}
static inline void hipemu_raw_buffer_store_b64(hipemu_u32x2v v, hipemu_rsrc r, unsigned voff, unsigned soff, int) {
  memcpy(r.base + voff + soff, &v, 8);
}
#define __builtin_amdgcn_raw_buffer_load_b64 hipemu_raw_buffer_load_b64
#define __builtin_amdgcn_raw_buffer_store_b64 hipemu_raw_buffer_store_b64
#define __builtin_amdgcn_raw_buffer_load_b32 hipemu_raw_buffer_load_b32
#define __builtin_amdgcn_raw_buffer_load_b128 hipemu_raw_buffer_load_b128
#define __builtin_amdgcn_raw_buffer_store_b128 hipemu_raw_buffer_store_b128
#define __builtin_amdgcn_s_sleep(n) hipemu::spin_yield()
#define __builtin_amdgcn_sched_barrier(m) ((void)0)

// ---- wave collectives (the builtins of the gfx950 target, by their ISA semantics) ------------------------------
typedef unsigned hipemu_u32x2 __attribute__((ext_vector_type(2)));
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 hipemu_bf16x8 __attribute__((ext_vector_type(8)));

static inline void hipemu_wave_barrier() { hipemu::wave_op(hipemu::OP_WAVE_BARRIER, 0, 0, 0, 0, 0, 0, 0, 0); }
static inline int hipemu_readfirstlane(int v) {
  int r; hipemu::wave_op(hipemu::OP_READFIRSTLANE, &v, 0, 0, &r, 0, 0, 0, 0); return r;
}
static inline int hipemu_readlane(int v, int lane) {
  int r; hipemu::wave_op(hipemu::OP_READLANE, &v, 0, 0, &r, lane, 0, 0, 0); return r;
}
static inline int hipemu_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  int r; hipemu::wave_op(hipemu::OP_DPP, &old, &src, 0, &r, ctrl, row_mask, bank_mask, bound_ctrl); return r;
}
static inline hipemu_u32x2 hipemu_permlane32_swap(unsigned old, unsigned src, bool, bool) {
  unsigned r[2]; hipemu::wave_op(hipemu::OP_PERMLANE32_SWAP, &old, &src, 0, r, 0, 0, 0, 0);
  hipemu_u32x2 v; v[0] = r[0]; v[1] = r[1]; return v;
}
static inline hipemu_u32x2 hipemu_permlane16_swap(unsigned old, unsigned src, bool, bool) {
  unsigned r[2]; hipemu::wave_op(hipemu::OP_PERMLANE16_SWAP, &old, &src, 0, r, 0, 0, 0, 0);
  hipemu_u32x2 v; v[0] = r[0]; v[1] = r[1]; return v;
}
static inline hipemu_f32x16 hipemu_mfma_f32_32x32x2f32(float a, float b, hipemu_f32x16 c, int, int, int) {
  float ci[16], co[16];
  for (int i = 0; i < 16; i++) ci[i] = c[i];
  hipemu::wave_op(hipemu::OP_MFMA_F32_32X32X2, &a, &b, ci, co, 0, 0, 0, 0);
  hipemu_f32x16 d;
  for (int i = 0; i < 16; i++) d[i] = co[i];
  return d;
}
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
static inline hipemu_f32x4 hipemu_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
  float ci[4], co[4];
  for (int i = 0; i < 4; i++) ci[i] = c[i];
  hipemu::wave_op(hipemu::OP_MFMA_F32_16X16X4, &a, &b, ci, co, 0, 0, 0, 0);
  hipemu_f32x4 d;
  for (int i = 0; i < 4; i++) d[i] = co[i];
  return d;
}
typedef _Float16 hipemu_f16x8 __attribute__((ext_vector_type(8)));
static inline hipemu_f32x16 hipemu_mfma_f32_32x32x16_f16(hipemu_f16x8 a, hipemu_f16x8 b, hipemu_f32x16 c, int, int, int) {
  float ai[8], bi[8], ci[16], co[16];                      // (same register layout as the bf16 variant)
  for (int i = 0; i < 8; i++) { ai[i] = (float)a[i]; bi[i] = (float)b[i]; }
  for (int i = 0; i < 16; i++) ci[i] = c[i];
  hipemu::wave_op(hipemu::OP_MFMA_BF16_32X32X16, ai, bi, ci, co, 0, 0, 0, 0);
  hipemu_f32x16 d;
  for (int i = 0; i < 16; i++) d[i] = co[i];
  return d;
}
static inline hipemu_f32x4 hipemu_mfma_f32_16x16x32_f16(hipemu_f16x8 a, hipemu_f16x8 b, hipemu_f32x4 c, int, int, int) {
  float ai[8], bi[8], ci[4], co[4];
  for (int i = 0; i < 8; i++) { ai[i] = (float)a[i]; bi[i] = (float)b[i]; }
  for (int i = 0; i < 4; i++) ci[i] = c[i];
  hipemu::wave_op(hipemu::OP_MFMA_F16_16X16X32, ai, bi, ci, co, 0, 0, 0, 0);
  hipemu_f32x4 d;
  for (int i = 0; i < 4; i++) d[i] = co[i];
  return d;
}
static inline hipemu_f32x16 hipemu_mfma_f32_32x32x16_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x16 c, int, int, int) {
  float ai[8], bi[8], ci[16], co[16];
  for (int i = 0; i < 8; i++) { ai[i] = (float)a[i]; bi[i] = (float)b[i]; }
  for (int i = 0; i < 16; i++) ci[i] = c[i];
  hipemu::wave_op(hipemu::OP_MFMA_BF16_32X32X16, ai, bi, ci, co, 0, 0, 0, 0);
  hipemu_f32x16 d;
  for (int i = 0; i < 16; i++) d[i] = co[i];
  return d;
}
// (64-bit values travel as two 32-bit halves, as the device's own __shfl of a double does)
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "hipemu: 32- and 64-bit shuffles only");
  int iv[2] = {0, 0}, r[2] = {0, 0}; memcpy(iv, &v, sizeof(T));
  for (unsigned h = 0; h < sizeof(T) / 4; h++) hipemu::wave_op(hipemu::OP_SHFL_XOR, &iv[h], 0, 0, &r[h], mask, width, 0, 0);
  T o; memcpy(&o, r, sizeof(T)); return o;
}
template <class T> static inline T __shfl(T v, int src, int width = 64) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "hipemu: 32- and 64-bit shuffles only");
  int iv[2] = {0, 0}, r[2] = {0, 0}; memcpy(iv, &v, sizeof(T));
  for (unsigned h = 0; h < sizeof(T) / 4; h++) hipemu::wave_op(hipemu::OP_SHFL, &iv[h], 0, 0, &r[h], src, width, 0, 0);
  T o; memcpy(&o, r, sizeof(T)); return o;
}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline unsigned long long __ballot(int pred) {
  unsigned long long r; hipemu::wave_op(hipemu::OP_BALLOT, &pred, 0, 0, &r, 0, 0, 0, 0); return r;
}
#define __builtin_amdgcn_wave_barrier hipemu_wave_barrier
#define __builtin_amdgcn_readfirstlane hipemu_readfirstlane
#define __builtin_amdgcn_readlane hipemu_readlane
#define __builtin_amdgcn_update_dpp hipemu_update_dpp
#define __builtin_amdgcn_permlane32_swap hipemu_permlane32_swap
#define __builtin_amdgcn_permlane16_swap hipemu_permlane16_swap
#define __builtin_amdgcn_mfma_f32_32x32x2f32 hipemu_mfma_f32_32x32x2f32
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu_mfma_f32_16x16x4f32
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 hipemu_mfma_f32_32x32x16_bf16
#define __builtin_amdgcn_mfma_f32_32x32x16_f16 hipemu_mfma_f32_32x32x16_f16
#define __builtin_amdgcn_mfma_f32_16x16x32_f16 hipemu_mfma_f32_16x16x32_f16
