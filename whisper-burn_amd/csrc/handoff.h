// Cross-workgroup hand-offs INSIDE one launch (the persistent decode kernel, decode_persist.hip).
//
// gfx950 has 8 XCDs with private, mutually non-coherent L2s, and a CU's vector L1 is never refreshed by another CU's
// stores.  The protocol used here is the write-through form of MI355X_MICROARCH.md "Workgroup dispatch, XCD placement &
// inter-workgroup visibility" / cdna_hip_programming.md section 6, Guideline 16 (R1), in its COUNTER form:
//
//   producer   payload with `sc1` (write-through, agent-scope) stores  ->  EVERY storing wave `s_waitcnt vmcnt(0)`  ->
//              __syncthreads()  ->  ONE lane: relaxed agent-scope atomic add on the consumers' arrival counter
//   consumer   ONE lane polls that ONE word with relaxed agent-scope loads (+ s_sleep)  ->  __syncthreads()  ->
//              payload with `sc1` loads (they bypass the L1; "sc1 loads may replace the acquire only when the producer
//              stored sc1")
//
// Counters are monotonic within a launch (target = (epoch + 1) x arrivals per epoch) and zeroed by the host before
// every launch.  Every spin is bounded: on give-up the error word is set and every block leaves the kernel.
// Results never depend on dispatch order, timing or block -> XCD placement.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wb {

// (the functional model tools/hipemu predefines these two hooks; everything else in this file is plain HIP)
#ifndef WB_DRAIN_VMEM
#define WB_DRAIN_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")   // inline asm: invisible to the waitcnt pass
#endif
#ifndef WB_LAUNCH_COOP
#define WB_LAUNCH_COOP(kernel, grid, block, shmem, stream, arg) \
  [&]() { void* _args[] = {(void*)&(arg)}; return hipLaunchCooperativeKernel((const void*)(kernel), grid, block, _args, shmem, stream); }()
#endif

// an ordering point for the compiler: the float4 is computed before it, no memory operation crosses it
#ifndef WB_PIN_F4
#define WB_PIN_F4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w) : : "memory")
#endif
// A role of the persistent kernel runs inside the step / role loops: without this, every thread-index-derived address of
// every role is loop-invariant, gets hoisted in front of the loops and lives (spilled) across all of them.
#ifndef WB_LAUNDER_V
#define WB_LAUNDER_V(x) asm volatile("" : "+v"(x))
#endif
template <bool PS>
__device__ __forceinline__ int role_tid() {
  int t = threadIdx.x;
  if constexpr (PS) WB_LAUNDER_V(t);
  return t;
}

typedef unsigned hx_u32x4 __attribute__((ext_vector_type(4)));

// ---- sc1 (agent-scope, L1-bypassing / write-through) accesses -----------------------------------------------------
template <bool SC1>
__device__ __forceinline__ float ld_f(const float* p) {
  if constexpr (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool SC1>
__device__ __forceinline__ int ld_i(const int* p) {
  if constexpr (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool SC1>
__device__ __forceinline__ void st_f(float* p, float v) {
  if constexpr (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
__device__ __forceinline__ void st_i_sc1(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#if defined(HIPEMU)
#define WB_GLOBAL_AS
#else
#define WB_GLOBAL_AS __attribute__((address_space(1)))
#endif
typedef float hx_f32x4 __attribute__((ext_vector_type(4)));
// 16-byte accesses go through a raw buffer resource (aux = 16 selects sc1 on gfx950); `base` must be 16-byte aligned
// and the byte offset < 4 GiB
struct Buf16 {
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ explicit Buf16(const void* base)
      : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000)) {}
};
template <bool SC1>
__device__ __forceinline__ float4 ld_f4(const float* base, const Buf16& b, uint32_t elem_off) {
  if constexpr (SC1) {
    const hx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b.r, elem_off * 4u, 0, 16);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
  } else {
    const hx_f32x4 v = *(const WB_GLOBAL_AS hx_f32x4*)(base + elem_off);
    return make_float4(v.x, v.y, v.z, v.w);
  }
}
// 16-byte load at element offset elem_soff (wave-uniform) + elem_voff, default cache policy.  (nt on the E^T tile stream and
// on the layer weights / cached K,V was measured in round 5: 13.66 -> 13.77 ms and -> 15.6 ms per bench step -- the weights
// are RE-read every step from the L2 / Infinity Cache, the guide's "nt-weights" row is about a stream read once.)
__device__ __forceinline__ float4 ld_f4_plain(const Buf16& b, uint32_t elem_voff, uint32_t elem_soff) {
  const hx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b.r, elem_voff * 4u, elem_soff * 4u, 0);
  return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}
// ---- loads through an EXPLICITLY global pointer ---------------------------------------------------------------------------
// The persistent kernel reads its per-layer argument blocks from device memory (PersistArgs::layers), so the compiler cannot
// know that the pointers inside them are global addresses: every weight / bias / cached-K,V access of the role bodies was a
// FLAT load (1 208 flat_load_dwordx4 + 144 flat_load_dword in the round-4 ISA of dec_persist_kernel).  A flat access counts
// on vmcnt AND lgkmcnt, so every wait for an LDS read (s_waitcnt lgkmcnt(0)) also waited for the weight rounds requested for
// LATER use -- the prefetch could not overlap the role's own LDS phases -- and every address was a 64-bit VGPR pair.  The
// role bodies therefore load global data through these helpers (global_load: vmcnt only).
__device__ __forceinline__ float4 gld4(const float* p) {
  const hx_f32x4 v = *(const WB_GLOBAL_AS hx_f32x4*)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float gld(const float* p) { return *(const WB_GLOBAL_AS float*)p; }
__device__ __forceinline__ int gld(const int* p) { return *(const WB_GLOBAL_AS int*)p; }
__device__ __forceinline__ float4 ld_w4(const float* p) { return gld4(p); }     // a 16-byte piece of a decoder weight matrix
// 4-byte load at base[elem_soff + elem_voff]: `elem_soff` must be wave-uniform (it rides in the instruction's scalar
// offset, so the per-lane offset register is shared by all planes of a fold)
template <bool SC1>
__device__ __forceinline__ float ld_fb(const float* base, const Buf16& b, uint32_t elem_voff, uint32_t elem_soff) {
  if constexpr (SC1) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b.r, elem_voff * 4u, elem_soff * 4u, 16));
  else return *(const WB_GLOBAL_AS float*)(base + (int64_t)elem_soff + elem_voff);
}
template <bool SC1>
__device__ __forceinline__ void st_f4(float* base, const Buf16& b, uint32_t elem_off, float4 v) {
  if constexpr (SC1) {
    hx_u32x4 u;
    u[0] = __float_as_uint(v.x); u[1] = __float_as_uint(v.y); u[2] = __float_as_uint(v.z); u[3] = __float_as_uint(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(u, b.r, elem_off * 4u, 0, 16);
  } else {
    *reinterpret_cast<float4*>(base + elem_off) = v;
  }
}

// ---- granules: the data IS the flag (Guideline 16, R2) ----------------------------------------------------------------
// A value another block waits for travels as an aligned 8-byte {tag, value} pair written by ONE write-through store (16-byte
// stores carry two pairs; observed untorn on gfx950).  The consumer re-reads its granules until every tag equals the tag
// of the producing stage: arrival and payload come back in the SAME round trip, where a flag costs one round trip to see
// the flag and a second one to fetch the data.  Tags are unique per (launch, step, layer, sublayer) and never 0; the
// buffers are zero-filled when allocated.
typedef unsigned hx_u32x2 __attribute__((ext_vector_type(2)));
struct Gran { unsigned tag; float v; };
__device__ __forceinline__ Gran ld_gran(const Buf16& b, uint32_t gran_voff, uint32_t gran_soff) {
  const hx_u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(b.r, gran_voff * 8u, gran_soff * 8u, 16);
  return Gran{u[0], __uint_as_float(u[1])};
}
__device__ __forceinline__ void st_gran(const Buf16& b, uint32_t gran_off, unsigned tag, float v) {
  hx_u32x2 u; u[0] = tag; u[1] = __float_as_uint(v);
  __builtin_amdgcn_raw_buffer_store_b64(u, b.r, gran_off * 8u, 0, 16);
}
__device__ __forceinline__ void st_gran4(const Buf16& b, uint32_t gran_off, unsigned tag, float4 v) {   // gran_off even
  hx_u32x4 u;
  u[0] = tag; u[1] = __float_as_uint(v.x); u[2] = tag; u[3] = __float_as_uint(v.y);
  __builtin_amdgcn_raw_buffer_store_b128(u, b.r, gran_off * 8u, 0, 16);
  u[1] = __float_as_uint(v.z); u[3] = __float_as_uint(v.w);
  __builtin_amdgcn_raw_buffer_store_b128(u, b.r, gran_off * 8u + 16u, 0, 16);
}
constexpr unsigned HX_SWEEP_LIMIT = 1u << 20;     // sweeps before a granule wait gives up

// ---- arrival counters ---------------------------------------------------------------------------------------------
constexpr unsigned HX_SPIN_LIMIT = 4u * 1000u * 1000u;      // polls before a wait gives up (seconds; a real wait lasts microseconds)

// Control words of one persistent launch (device memory, zeroed / initialised by the host before the launch).  Every
// polled word owns a 128-byte line (HX_LINE ints apart): a few hundred pollers and the producers' atomics on ONE line
// serialise at its memory channel (measured: 7-9 us per hand-off with all counters in one line, r03_c_ps_timeline_first).
constexpr int HX_LINE = 32;
enum { HX_STOP = 0,               // first chain step that must NOT run (INT_MAX while the decode goes on)
       HX_ERR = HX_LINE,          // != 0: a wait gave up (value = 1 + index of the counter) -- every block leaves
       HX_NDONE = 2 * HX_LINE,    // rows finished so far
       HX_HDR = 3 * HX_LINE };    // counters start here, counter c at HX_HDR + c * HX_LINE

// Publish: call with ALL threads of the block after the role's last payload store.
__device__ __forceinline__ void hx_arrive(unsigned* ctr) {
  WB_DRAIN_VMEM();                     // every storing wave: its write-through stores have left the CU
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wait until *ctr >= target.  Returns false when the decode was stopped at or before `step` (every window has ended)
// or a wait anywhere gave up: the caller leaves the kernel without arriving anywhere.  Call with ALL threads.
__device__ __forceinline__ bool hx_wait(const unsigned* ctr, unsigned target, const int* ctl, int step, int ctr_index,
                                        int* lds_flag) {
  if (threadIdx.x == 0) {
    int ok = 1;
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      // (the stop / error words are everybody's: looked at every 16th poll only)
      if ((spins & 15u) == 15u &&
          (__hip_atomic_load(ctl + HX_STOP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= step ||
           __hip_atomic_load(ctl + HX_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { ok = 0; break; }
      if (++spins > HX_SPIN_LIMIT) {
        __hip_atomic_store(const_cast<int*>(ctl) + HX_ERR, 1 + ctr_index, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    if (ok && __hip_atomic_load(ctl + HX_STOP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= step) ok = 0;
    *lds_flag = ok;
  }
  __syncthreads();
  const int ok = *lds_flag;
  __syncthreads();                     // (the flag word is reused by the next wait)
  return ok != 0;
}

// Wait until ctr[k HX_LINE] >= target[k] for every k < n (n <= 64): lane k of wave 0 polls counter k, so that the wait costs
// ONE detection latency instead of n.  Same contract as hx_wait.
__device__ __forceinline__ bool hx_wait_many(const unsigned* ctr, const unsigned* target, int n, const int* ctl, int step,
                                             int ctr_index, int* lds_flag) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const bool mine = lane < n;
    const unsigned tgt = mine ? target[lane] : 0u;
    int ok = 1;
    unsigned spins = 0;
    for (;;) {
      const bool have = !mine || __hip_atomic_load(ctr + lane * HX_LINE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= tgt;
      if (__ballot(!have) == 0ull) break;
      if ((spins & 15u) == 15u &&
          (__hip_atomic_load(ctl + HX_STOP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= step ||
           __hip_atomic_load(ctl + HX_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { ok = 0; break; }
      if (++spins > HX_SPIN_LIMIT) {
        if (lane == 0) __hip_atomic_store(const_cast<int*>(ctl) + HX_ERR, 1 + ctr_index, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    if (ok && __hip_atomic_load(ctl + HX_STOP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= step) ok = 0;
    if (lane == 0) *lds_flag = ok;
  }
  __syncthreads();
  const int ok = *lds_flag;
  __syncthreads();
  return ok != 0;
}

// The same for n consecutive counters that share one target.
__device__ __forceinline__ bool hx_wait_same(const unsigned* ctr, unsigned target, int n, const int* ctl, int step,
                                             int ctr_index, int* lds_flag) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const bool mine = lane < n;
    int ok = 1;
    unsigned spins = 0;
    for (;;) {
      const bool have = !mine || __hip_atomic_load(ctr + lane * HX_LINE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target;
      if (__ballot(!have) == 0ull) break;
      if ((spins & 15u) == 15u &&
          (__hip_atomic_load(ctl + HX_STOP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= step ||
           __hip_atomic_load(ctl + HX_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { ok = 0; break; }
      if (++spins > HX_SPIN_LIMIT) {
        if (lane == 0) __hip_atomic_store(const_cast<int*>(ctl) + HX_ERR, 1 + ctr_index, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    if (ok && __hip_atomic_load(ctl + HX_STOP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= step) ok = 0;
    if (lane == 0) *lds_flag = ok;
  }
  __syncthreads();
  const int ok = *lds_flag;
  __syncthreads();
  return ok != 0;
}

// Arrive, and if this block is the LAST of the epoch's `total` arrivals (the counter reaches `full`), wake the many
// consumers of this stage through `n_go` words on lines of their own: hundreds of blocks polling ONE word see a new
// value only after the whole queue of polls in front of it has been served (measured ~6 us for 203 pollers).  Every
// earlier arriver drained its stores before its add, so the payload of all of them is in memory when the words change.
__device__ __forceinline__ void hx_arrive_broadcast(unsigned* ctr, unsigned full, unsigned* go, int n_go, unsigned epoch) {
  WB_DRAIN_VMEM();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u == full)
      for (int k = 0; k < n_go; k++) __hip_atomic_store(go + k * HX_LINE, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace wb
