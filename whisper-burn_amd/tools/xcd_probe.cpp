// Developer probe (not part of the library): what does a {tag, value} granule hand-off cost between two co-resident
// blocks on the SAME XCD against two blocks on DIFFERENT XCDs, by cache policy of the store / load?  The persistent decode
// kernel (decode_persist.hip) hands every plane over with write-through `sc1` stores and L1-bypassing `sc1` loads because
// the 8 XCD L2s are not coherent with each other: ~1.2 - 2 us per hand-off, 14 of them per token.  If blocks of one XCD can
// exchange through their own L2 (sc0: L1 bypass only), the intra-layer hand-offs (attention -> cross-attention -> MLP) could
// stay inside an XCD when the host deals a layer's roles to one XCD.  This probe measures exactly that:
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o xcd_probe xcd_probe.cpp && ./xcd_probe
//
// 1. 256 co-resident blocks record their XCC_ID: is block b on XCD b % 8 (the dealing rule would rely on it)?
// 2. ping-pong between block pairs (partner = b + 8: same XCD, or b + 1: next XCD), 2000 round trips each, for the policies
//    aux = 0 (default), 1 (sc0), 16 (sc1), 17 (sc0 | sc1) on both the store and the load.  Reports the median round trip in
//    ns (one round trip = two hand-offs) or "never seen" when the partner's store did not become visible within the bound.
// Every spin is bounded (a probe must never hang the box).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7fffffff, 0x00020000);
}

__global__ void xcc_kernel(int* out) {
  if (threadIdx.x == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    out[blockIdx.x] = (int)(id & 0xf);
  }
}

template <int AUX>
__global__ __launch_bounds__(64) void pingpong_kernel(void* slots, int stride_blocks, int iters, unsigned long long* ns_out,
                                                      int* fail) {
  // blocks pair up: b and b + stride_blocks (b in the lower half of each 2 * stride group); slot[b] is written by b's
  // partner and read by b.  Only lane 0 works: this is a latency probe.
  const int b = blockIdx.x;
  const int group = b / (2 * stride_blocks), pos = b % (2 * stride_blocks);
  const bool lead = pos < stride_blocks;
  const int partner = lead ? b + stride_blocks : b - stride_blocks;
  if (partner >= (int)gridDim.x) return;
  if (threadIdx.x != 0) return;
  (void)group;
  const __amdgpu_buffer_rsrc_t r = rsrc(slots);
  const unsigned mine = (unsigned)b * 32u * 8u, theirs = (unsigned)partner * 32u * 8u;   // one 256-byte line per block
  unsigned long long t0 = 0;
  bool ok = true;
  for (int i = 1; i <= iters && ok; i++) {
    if (lead) {
      if (i == 11) t0 = wall_clock64();                // (ten round trips of warm-up)
      u32x2 v; v[0] = (unsigned)i; v[1] = (unsigned)i * 2654435761u;
      __builtin_amdgcn_raw_buffer_store_b64(v, r, theirs, 0, AUX);
    }
    // wait for the partner's i
    unsigned spins = 0;
    for (;;) {
      const u32x2 g = __builtin_amdgcn_raw_buffer_load_b64(r, mine, 0, AUX);
      if (g[0] == (unsigned)i) { if (g[1] != (unsigned)i * 2654435761u) { atomicAdd(fail, 1000000); } break; }
      if (++spins > 2000000u) { ok = false; break; }
    }
    if (!lead && ok) {
      u32x2 v; v[0] = (unsigned)i; v[1] = (unsigned)i * 2654435761u;
      __builtin_amdgcn_raw_buffer_store_b64(v, r, theirs, 0, AUX);
    }
  }
  if (!ok) atomicAdd(fail, 1);
  if (lead) ns_out[b] = ok ? (wall_clock64() - t0) * 10ull : 0ull;      // wall_clock64: 100 MHz
}

template <int AUX>
static void run(const char* name, void* slots, int stride, unsigned long long* ns_dev, int* fail_dev, int nb) {
  const int iters = 2010;
  CHECK(hipMemset(slots, 0, (size_t)nb * 256));
  CHECK(hipMemset(ns_dev, 0, (size_t)nb * 8));
  CHECK(hipMemset(fail_dev, 0, 4));
  int it = iters;
  void* args[] = {&slots, &stride, &it, &ns_dev, &fail_dev};
  CHECK(hipLaunchCooperativeKernel((const void*)pingpong_kernel<AUX>, dim3(nb), dim3(64), args, 0, nullptr));
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned long long> ns(nb);
  int fail = 0;
  CHECK(hipMemcpy(ns.data(), ns_dev, (size_t)nb * 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(&fail, fail_dev, 4, hipMemcpyDeviceToHost));
  std::vector<double> rt;
  for (int b = 0; b < nb; b++)
    if (ns[b]) rt.push_back((double)ns[b] / 2000.0);
  std::sort(rt.begin(), rt.end());
  if (rt.empty()) { printf("  %-22s partner +%d: never seen (%d pairs gave up)\n", name, stride, fail % 1000000); return; }
  printf("  %-22s partner +%d: round trip median %7.0f ns  min %7.0f  max %7.0f  (%zu pairs; gave up %d, torn %d)\n", name, stride,
         rt[rt.size() / 2], rt.front(), rt.back(), rt.size(), fail % 1000000, fail / 1000000);
}

int main() {
  CHECK(hipSetDevice(0));
  const int nb = 256;
  int* xcc;
  CHECK(hipMalloc((void**)&xcc, nb * 4));
  void* a0[] = {&xcc};
  CHECK(hipLaunchCooperativeKernel((const void*)xcc_kernel, dim3(nb), dim3(64), a0, 0, nullptr));
  CHECK(hipDeviceSynchronize());
  std::vector<int> h(nb);
  CHECK(hipMemcpy(h.data(), xcc, nb * 4, hipMemcpyDeviceToHost));
  int match = 0;
  for (int b = 0; b < nb; b++) match += (h[b] == b % 8);
  printf("XCC_ID of block b == b %% 8 for %d of %d blocks; first 16: ", match, nb);
  for (int b = 0; b < 16; b++) printf("%d ", h[b]);
  printf("\n");
  void* slots; unsigned long long* ns; int* fail;
  CHECK(hipMalloc(&slots, (size_t)nb * 256));
  CHECK(hipMalloc((void**)&ns, (size_t)nb * 8));
  CHECK(hipMalloc((void**)&fail, 4));
  for (int stride : {8, 1}) {
    printf("partner = block + %d (%s):\n", stride, stride == 8 ? "same XCD if b %% 8 is the XCD" : "the next XCD");
    run<0>("aux 0 (default)", slots, stride, ns, fail, nb);
    run<1>("aux 1 (sc0)", slots, stride, ns, fail, nb);
    run<16>("aux 16 (sc1)", slots, stride, ns, fail, nb);
    run<17>("aux 17 (sc0 | sc1)", slots, stride, ns, fail, nb);
  }
  return 0;
}
