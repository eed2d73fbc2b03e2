// Developer probe (not part of the library): what does a grid-wide barrier INSIDE a co-resident kernel cost on MI355X, next
// to the ~8.5 us of idle time behind every launch of the batch-mode decode chain (profiles/r03_m_layer_cycle_large_v2.txt)?
// The batch-mode counterpart of the persistent decode kernel (DESIGN.md "what comes next", 2) stands or falls with this number.
//
//   hipcc --offload-arch=gfx950 -O3 -o gridsync_probe gridsync_probe.cpp && ./gridsync_probe
//
// 256 blocks x 512 threads (one per CU, launched co-operatively), 2000 barriers each:
//   flat        one agent-scope atomic add per block on ONE word, everybody polls that word
//   flat+fence  the same between an agent-scope release fence and an acquire fence (L2 write-back / invalidate: ordinary
//               loads and stores of the phases become visible across XCDs -- what a kernel boundary does)
//   xcd         arrival per XCD first (block % 8), the last arriver of an XCD adds to the global word; everybody polls it
//   xcd+fence   ... with the fences
//   payload     flat + write-through (sc1) hand-off of 4 KB per block read by the next block (Guideline 16 R1, no fences)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NT = 512;
constexpr int LINE = 32;   // ints per 128-byte line

// bounded: a probe must never hang the box -- on give-up the word is pushed past every target and everybody leaves
__device__ __forceinline__ bool spin_until(unsigned* p, unsigned target) {
  for (unsigned n = 0; __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; n++) {
    if (n > 20000000u) { __hip_atomic_store(p, 0x7fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
    __builtin_amdgcn_s_sleep(1);
  }
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0x7fffffffu;
}

template <int MODE>   // 0 flat, 1 flat+fence, 2 xcd, 3 xcd+fence, 4 payload
__global__ __launch_bounds__(NT) void probe_kernel(unsigned* ctr, float* buf, int iters, float* sink) {
  const int nb = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
  unsigned* global = ctr;                       // line 0
  unsigned* xcd = ctr + LINE * (1 + (b & 7));   // lines 1..8
  const int per_xcd = (nb + 7 - (b & 7)) / 8;   // blocks with this residue
  float acc = 0.f;
  __shared__ int alive;
  if (tid == 0) alive = 1;
  __syncthreads();
  for (int it = 0; it < iters; it++) {
    if (MODE == 4) {
      // 4 KB per block, write-through, every wave drained before the arrival
      float* mine = buf + (size_t)b * 1024;
      __hip_atomic_store(mine + tid, (float)(it + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mine + 512 + tid, (float)(it - tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (MODE == 1 || MODE == 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) {
      if (MODE == 2 || MODE == 3) {
        const unsigned t = __hip_atomic_fetch_add(xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (unsigned)(it + 1) * per_xcd - 1) __hip_atomic_fetch_add(global, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!spin_until(global, (unsigned)(it + 1) * 8u)) alive = 0;
      } else {
        __hip_atomic_fetch_add(global, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!spin_until(global, (unsigned)(it + 1) * nb)) alive = 0;
      }
    }
    __syncthreads();
    if (!alive) { if (tid == 0) sink[1] = 1.f; return; }
    if (MODE == 1 || MODE == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (MODE == 4) {
      const float* other = buf + (size_t)((b + 1) % nb) * 1024;
      acc += __hip_atomic_load(other + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
             __hip_atomic_load(other + 512 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (acc == 12345.f) sink[0] = acc;
}

template <int MODE>
static void run(const char* name, unsigned* ctr, float* buf, float* sink, int nb, int iters) {
  CHECK(hipMemset(ctr, 0, LINE * 9 * sizeof(unsigned)));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  void* args[] = {&ctr, &buf, &iters, &sink};
  CHECK(hipEventRecord(a, 0));
  CHECK(hipLaunchCooperativeKernel((const void*)probe_kernel<MODE>, dim3(nb), dim3(NT), args, 0, 0));
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, a, b));
  float flag[2] = {0.f, 0.f};
  CHECK(hipMemcpy(flag, sink, 8, hipMemcpyDeviceToHost));
  printf("%-12s %4d blocks  %6.2f us per barrier%s\n", name, nb, ms * 1e3f / iters, flag[1] != 0.f ? "  (GAVE UP)" : "");
}

int main() {
  int cus = 0;
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  unsigned* ctr; float* buf; float* sink;
  CHECK(hipMalloc(&ctr, LINE * 9 * sizeof(unsigned)));
  CHECK(hipMalloc(&buf, (size_t)cus * 1024 * sizeof(float)));
  CHECK(hipMalloc(&sink, 8));
  CHECK(hipMemset(sink, 0, 8));
  const int iters = 2000;
  for (int rep = 0; rep < 2; rep++) {
    run<0>("flat", ctr, buf, sink, cus, iters);
    run<1>("flat+fence", ctr, buf, sink, cus, iters);
    run<2>("xcd", ctr, buf, sink, cus, iters);
    run<3>("xcd+fence", ctr, buf, sink, cus, iters);
    run<4>("payload", ctr, buf, sink, cus, iters);
  }
  return 0;
}
