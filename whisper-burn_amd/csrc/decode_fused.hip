// Fused small-batch decode-step kernels: one launch per sublayer instead of one per matrix.
//
// With a handful of live beams a decode step is bound by its chain of DEPENDENT launches, not by bytes
// (tiny.en: 30 launches of 4-6 us around 17 us of streaming).  Two producer-aligned fusions remove kernel
// boundaries without any grid-wide synchronisation, because the block that OWNS a slice of an intermediate
// can apply the next matrix to it on its own and leave a partial plane that the consumer's prologue already
// knows how to fold (fixed order, no atomics -> bit-reproducible):
//
//   dec_mlp_fused_kernel    block j owns hidden units [64 j, 64 j + 64):  LN(x + pending) -> W1[:, slice]
//                           -> + b1 -> GELU -> W2[slice, :]  -> plane j of [4 d / 64][S][d]   (mod.rs:376-382)
//   dec_attn_fused_kernel   block h owns head h:  LN(x + pending) -> Wqkv[:, head h] -> cache append -> masked
//                           self-attention over the paged cache -> Wo[head h rows, :] -> plane h of [H][S][d]
//                                                                            (mod.rs:428-436, :493-533)
//
// Every block redoes the row prologue (fold + LayerNorm: a few KB from L2); each weight is still read once.
#include <hip/hip_runtime.h>

#include "decode.h"
#include "decode_fused_bodies.h"
#include "wave_ops.h"

namespace wb {
namespace {

using namespace fused;

// one launch per sublayer: grid = the roles of that sublayer, the step state comes from a.st
template <int MR, int DPL, bool REC>
__global__ __launch_bounds__(512) void dec_mlp_fused_kernel(MlpFusedArgs a) {
  if (blockIdx.y > 0) {                            // row group g: rows [g MR, g MR + MR) (9 - 16 live rows; plain planes only)
    const int r0 = blockIdx.y * MR;
    const int64_t sh = (int64_t)r0 * a.d;
    a.row0 = r0; a.x_in += sh; a.x_out += sh; a.P += sh;
    if (a.pend) a.pend += sh;
  }
  dec_mlp_body<MR, DPL, REC, false>(a, blockIdx.x, PsStep());
}
// grid = (8, rows): workgroups go round-robin over the 8 XCDs by linear id, so x = head puts every beam's block of
// one head on the SAME XCD -- the head's weight slices cross the fabric once and are shared through that L2
template <int DPL>
__global__ __launch_bounds__(512) void dec_attn_fused_kernel(AttnFusedArgs a) {
  dec_attn_body<DPL, false>(a, blockIdx.x, blockIdx.y, PsStep());
}
template <int DPL, int NP>
__global__ __launch_bounds__(512) void dec_cross_fused_kernel(CrossFusedArgs a) {
  dec_cross_body<DPL, false, NP>(a, blockIdx.x, blockIdx.y, PsStep());
}

}  // namespace

bool dec_fused_supported(int d) { return d == 128 || d == 384 || d == 512; }
int dec_mlp_fused_planes(int d) { return 4 * d / 64; }

void launch_dec_mlp_fused(hipStream_t st, const MlpFusedArgs& a, int n_rows_hint) {
  // more than 8 rows: row groups of 8 in grid.y (the caller guarantees plain planes: n_chunks == 0)
  const dim3 grid(4 * a.d / 64, n_rows_hint > 8 ? (n_rows_hint + 7) / 8 : 1), block(512);
#define WB_MLP(MR_, DPL_)                                                                          \
  do {                                                                                             \
    if (a.n_chunks > 0) WB_KLAUNCH((dec_mlp_fused_kernel<MR_, DPL_, true>), grid, block, 0, st, a); \
    else WB_KLAUNCH((dec_mlp_fused_kernel<MR_, DPL_, false>), grid, block, 0, st, a);               \
  } while (0)
  if (n_rows_hint <= 4) {
    if (a.d == 128) WB_MLP(4, 2); else if (a.d == 384) WB_MLP(4, 6); else WB_MLP(4, 8);
  } else {
    if (a.d == 128) WB_MLP(8, 2); else if (a.d == 384) WB_MLP(8, 6); else WB_MLP(8, 8);
  }
#undef WB_MLP
}

void launch_dec_cross_fused(hipStream_t st, const CrossFusedArgs& a, int n_rows_hint) {
  const dim3 grid(a.n_head <= 8 ? 8 : a.n_head, n_rows_hint), block(512);
  // (a.n_pass = 2: some window holds more than CROSS_FUSED_MAX_C keys -- the two-pass ring)
  if (a.n_pass <= 1) {
    if (a.d == 128) WB_KLAUNCH((dec_cross_fused_kernel<2, 1>), grid, block, 0, st, a);
    else if (a.d == 384) WB_KLAUNCH((dec_cross_fused_kernel<6, 1>), grid, block, 0, st, a);
    else WB_KLAUNCH((dec_cross_fused_kernel<8, 1>), grid, block, 0, st, a);
  } else {
    if (a.d == 128) WB_KLAUNCH((dec_cross_fused_kernel<2, 2>), grid, block, 0, st, a);
    else if (a.d == 384) WB_KLAUNCH((dec_cross_fused_kernel<6, 2>), grid, block, 0, st, a);
    else WB_KLAUNCH((dec_cross_fused_kernel<8, 2>), grid, block, 0, st, a);
  }
}

void launch_dec_attn_fused(hipStream_t st, const AttnFusedArgs& a, int n_rows_hint) {
  const dim3 grid(a.n_head <= 8 ? 8 : a.n_head, n_rows_hint), block(512);
  if (a.d == 128) WB_KLAUNCH((dec_attn_fused_kernel<2>), grid, block, 0, st, a);
  else if (a.d == 384) WB_KLAUNCH((dec_attn_fused_kernel<6>), grid, block, 0, st, a);
  else WB_KLAUNCH((dec_attn_fused_kernel<8>), grid, block, 0, st, a);
}

}  // namespace wb
