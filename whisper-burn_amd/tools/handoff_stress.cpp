// GPU stress test of the hand-off granules of the persistent decode kernel (csrc/handoff.h): is an aligned 8-byte
// {tag, value} pair -- written by ONE write-through (sc1) 8-byte store, or two of them by ONE 16-byte store -- ever seen
// TORN (new tag, old value or the reverse) by an L1-bypassing load on another XCD?  The protocol of decode_persist.hip
// stands on "never": a consumer accepts a value as soon as its tag is the producing stage's.
//
//   make -C whisper-burn_amd/csrc stress && whisper-burn_amd/lib/handoff_stress [million_reads_per_mode] [device]
//
// 256 co-resident blocks (one per CU: the grid of the persistent kernel) x 256 threads.  Block b < 128 is a PRODUCER that
// keeps rewriting its 256 x 2 granules with (tag = n, value = mix(n, slot)) for n = 1, 2, ... as fast as it can; block
// b + 128 + 1 (another XCD: consecutive blocks go to consecutive XCDs) is its CONSUMER and keeps reading them, checking for
// every granule read that value == mix(tag, slot) and that tags never go backwards.  Producers do not wait for consumers:
// the granules change under the readers all the time, which is the worst case for tearing.  Modes:
//   0  8-byte store (st_gran)            read by 8-byte loads (ld_gran)           -- planes of the roles
//   1  16-byte store (st_gran4: two pairs) read by 8-byte loads                     -- residual streams (float4 producers)
//   2  16-byte store                       read by ONE 16-byte load, each pair checked on its own AND the two tags of a
//      16-byte store compared (informational: the protocol never relies on two pairs arriving together)
// Prints one JSON line per mode: reads, distinct tag changes seen, torn, backwards, pair_mismatch (mode 2).  Exit code 1
// when any granule was torn or went backwards.  Every loop is bounded (a probe must never hang the box).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "../csrc/handoff.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

using namespace wb;

constexpr int NT = 256;
constexpr int PAIRS = 128;                 // producer / consumer pairs

__device__ __forceinline__ unsigned mix(unsigned tag, unsigned slot) {
  unsigned x = tag * 2654435761u ^ (slot * 40503u + 0x9e3779b9u);
  x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
  return x & 0x7f7fffffu;                  // a finite float pattern (the value travels through float registers)
}

struct Counts { unsigned long long reads, changes, torn, backwards, pair_mismatch; };

template <int MODE>
__global__ __launch_bounds__(NT) void stress_kernel(void* gran, unsigned* stop, Counts* out, unsigned n_reads, int nap) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const bool producer = b < PAIRS;
  // consumer block c = PAIRS + p reads pair (p + PAIRS - 1) % PAIRS: producer and consumer sit on different XCDs
  const int pair = producer ? b : (b - PAIRS + PAIRS - 1) % PAIRS;
  const Buf16 buf(static_cast<char*>(gran) + (size_t)pair * NT * 2 * 8);
  const unsigned slot0 = 2u * tid;         // this thread's two granules (one 16-byte line piece)
  if (producer) {
    for (unsigned n = 1; n < 0x7fffff00u; n++) {
      if constexpr (MODE == 0) {
        st_gran(buf, slot0, n, __uint_as_float(mix(n, slot0)));
        st_gran(buf, slot0 + 1, n, __uint_as_float(mix(n, slot0 + 1)));
      } else {
        // st_gran4 writes FOUR granules with two 16-byte stores; here one 16-byte store = two granules per thread
        hx_u32x4 u;
        u[0] = n; u[1] = mix(n, slot0); u[2] = n; u[3] = mix(n, slot0 + 1);
        __builtin_amdgcn_raw_buffer_store_b128(u, buf.r, slot0 * 8u, 0, 16);
      }
      if ((n & 63u) == 0 && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)PAIRS) break;
    }
    return;
  }
  Counts c{0, 0, 0, 0, 0};
  unsigned last0 = 0, last1 = 0;
  for (unsigned i = 0; i < n_reads; i++) {
    unsigned t0, v0, t1, v1;
    if constexpr (MODE == 2) {
      const hx_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(buf.r, slot0 * 8u, 0, 16);
      t0 = u[0]; v0 = u[1]; t1 = u[2]; v1 = u[3];
      c.pair_mismatch += (t0 != t1);
    } else {
      const Gran g0 = ld_gran(buf, slot0, 0), g1 = ld_gran(buf, slot0 + 1, 0);
      t0 = g0.tag; v0 = __float_as_uint(g0.v); t1 = g1.tag; v1 = __float_as_uint(g1.v);
    }
    c.reads += 2;
    // throttled passes: flat out, the 32 768 readers pull ~4 TB/s of 8-byte granules and the writers' write-through stores
    // hardly land (a slot changes a handful of times per pass: profiles/r04_d_handoff_stress.txt); with a nap between reads the
    // writers get the fabric and the readers see far more distinct values per read
    if (nap) __builtin_amdgcn_s_sleep(32);
    if (t0 != 0) { c.torn += (v0 != mix(t0, slot0)); c.backwards += (t0 < last0); c.changes += (t0 != last0); last0 = t0; }
    if (t1 != 0) { c.torn += (v1 != mix(t1, slot0 + 1)); c.backwards += (t1 < last1); c.changes += (t1 != last1); last1 = t1; }
  }
  // block totals -> one atomic per counter per block
  __shared__ unsigned long long sh[5];
  if (tid < 5) sh[tid] = 0;
  __syncthreads();
  atomicAdd(&sh[0], c.reads); atomicAdd(&sh[1], c.changes); atomicAdd(&sh[2], c.torn); atomicAdd(&sh[3], c.backwards);
  atomicAdd(&sh[4], c.pair_mismatch);
  __syncthreads();
  if (tid == 0) {
    atomicAdd(&out->reads, sh[0]); atomicAdd(&out->changes, sh[1]); atomicAdd(&out->torn, sh[2]);
    atomicAdd(&out->backwards, sh[3]); atomicAdd(&out->pair_mismatch, sh[4]);
    __hip_atomic_fetch_add(stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // the producers leave when every consumer is done
  }
}

template <int MODE>
static bool run_mode(unsigned n_reads, void* gran, unsigned* stop, Counts* out_dev, int nap) {
  CHECK(hipMemset(gran, 0, (size_t)PAIRS * NT * 2 * 8));
  CHECK(hipMemset(stop, 0, 4));
  CHECK(hipMemset(out_dev, 0, sizeof(Counts)));
  void* args[] = {&gran, &stop, &out_dev, &n_reads, &nap};
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  CHECK(hipLaunchCooperativeKernel((const void*)stress_kernel<MODE>, dim3(2 * PAIRS), dim3(NT), args, 0, nullptr));
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  Counts c;
  CHECK(hipMemcpy(&c, out_dev, sizeof(c), hipMemcpyDeviceToHost));
  static const char* names[] = {"8B store / 8B load", "16B store / 8B loads", "16B store / 16B load"};
  printf("{\"mode\": \"%s\", \"readers\": \"%s\", \"granules_read\": %llu, \"tag_changes_seen\": %llu, \"torn\": %llu, \"backwards\": %llu, "
         "\"pair_mismatch\": %llu, \"ms\": %.1f}\n", names[MODE], nap ? "napping" : "flat out", c.reads, c.changes, c.torn, c.backwards, c.pair_mismatch, ms);
  return c.torn == 0 && c.backwards == 0 && c.changes > 0;
}

int main(int argc, char** argv) {
  const double mreads = argc > 1 ? atof(argv[1]) : 1100.0;        // million granules read per mode (default: > 1e9)
  const int device = argc > 2 ? atoi(argv[2]) : 0;
  CHECK(hipSetDevice(device));
  int coop = 0, cus = 0;
  CHECK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, device));
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
  if (!coop || cus < 2 * PAIRS) { printf("needs cooperative launches and %d CUs (have %d)\n", 2 * PAIRS, cus); return 2; }
  const unsigned n_reads = (unsigned)(mreads * 1e6 / (2.0 * PAIRS * NT)) + 1;
  void* gran; unsigned* stop; Counts* out;
  CHECK(hipMalloc(&gran, (size_t)PAIRS * NT * 2 * 8));
  CHECK(hipMalloc((void**)&stop, 4));
  CHECK(hipMalloc((void**)&out, sizeof(Counts)));
  bool ok = run_mode<0>(n_reads, gran, stop, out, 0);
  ok = run_mode<1>(n_reads, gran, stop, out, 0) && ok;
  ok = run_mode<2>(n_reads, gran, stop, out, 0) && ok;
  const unsigned n_nap = n_reads / 64 + 1;          // (a napping read takes ~1 us: 1/64 of the reads keeps the pass at ~0.1 - 0.3 s)
  ok = run_mode<0>(n_nap, gran, stop, out, 1) && ok;
  ok = run_mode<1>(n_nap, gran, stop, out, 1) && ok;
  ok = run_mode<2>(n_nap, gran, stop, out, 1) && ok;
  printf("%s\n", ok ? "HANDOFF_STRESS_OK" : "HANDOFF_STRESS_FAILED");
  return ok ? 0 : 1;
}
