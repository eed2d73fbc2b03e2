// Multi-GPU long-audio transcription behind the C ABI: the reference's windows sharded over ranks, ONE all-gather of the
// token rows, host stitch on every rank (include/whisper_hip.h "multi-GPU").
//
// Windows are independent in the reference -- the previous-window prompt is computed (transcribe.rs:43-50) and then
// discarded by a shadowing Vec::new() (transcribe.rs:195-201); only the token-overlap stitch (transcribe.rs:56-63) is
// sequential -- so rank r of R decodes the contiguous block [ceil(r K / R), ceil((r + 1) K / R)) of the K windows with
// no data-path collective; the only exchange is a fixed-shape int32 buffer [ceil(K / R)][1 + row_stride] per rank
// (length + tokens).  The exchange is a caller-supplied all-gather (any transport: the Python binding hands in
// torch.distributed's), or the built-in RCCL one (wb_comm_*: librccl is opened at run time, so the library itself carries
// no link-time dependency on it and single-GPU deployments never load it).
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "wb_internal.h"

using namespace wb;

// ---- the RCCL transport (opened lazily) ----------------------------------------------------------------------------
namespace {

struct NcclUniqueId { char internal[128]; };
typedef void* ncclComm_t;
enum { NCCL_INT32 = 2 };                          // ncclDataType_t: ncclInt8 0, ncclUint8 1, ncclInt32 2 (rccl.h)

struct Rccl {
  void* h = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

int rccl_load(Rccl** out) {
  static Rccl r;
  static int state = 0;                           // 0 untried, 1 ok, -1 failed
  if (state == 0) {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.h) break;
    }
    state = -1;
    if (r.h) {
      r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.h, "ncclGetUniqueId"));
      r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.h, "ncclCommInitRank"));
      r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.h, "ncclCommDestroy"));
      r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.h, "ncclAllGather"));
      r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.h, "ncclGetErrorString"));
      if (r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GetErrorString) state = 1;
    }
  }
  WB_REQUIRE(state == 1, WB_ERR_STATE, "RCCL is not available (librccl.so could not be opened): %s", r.h ? "symbols missing" : dlerror());
  *out = &r;
  return WB_OK;
}

}  // namespace

struct wb_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t st = nullptr;
  DevMem send, recv;
};

#define WB_NCCL(r, call)                                                                              \
  do {                                                                                                \
    const int rc_ = (call);                                                                           \
    if (rc_ != 0) { set_error("RCCL: %s failed: %s", #call, (r)->GetErrorString(rc_)); return WB_ERR_HIP; } \
  } while (0)

extern "C" {

int wb_comm_unique_id(uint8_t* id128) {
  WB_REQUIRE(id128, WB_ERR_ARG, "wb_comm_unique_id: null argument");
  Rccl* r;
  WB_TRY(rccl_load(&r));
  NcclUniqueId id;
  WB_NCCL(r, r->GetUniqueId(&id));
  memcpy(id128, id.internal, 128);
  return WB_OK;
}

int wb_comm_init(const uint8_t* id128, int rank, int world, int device, wb_comm** out) {
  WB_REQUIRE(id128 && out && world >= 1 && rank >= 0 && rank < world, WB_ERR_ARG, "wb_comm_init: bad argument");
  Rccl* r;
  WB_TRY(rccl_load(&r));
  WB_HIP(hipSetDevice(device));
  auto c = std::make_unique<wb_comm>();
  c->rank = rank; c->world = world; c->device = device;
  NcclUniqueId id;
  memcpy(id.internal, id128, 128);
  WB_NCCL(r, r->CommInitRank(&c->comm, world, id, rank));
  WB_HIP(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking));
  *out = c.release();
  return WB_OK;
}

void wb_comm_free(wb_comm* c) {
  if (!c) return;
  Rccl* r;
  (void)hipSetDevice(c->device);
  if (c->comm && rccl_load(&r) == WB_OK) (void)r->CommDestroy(c->comm);
  if (c->st) (void)hipStreamDestroy(c->st);
  delete c;
}

// wb_allgather_fn over RCCL: `user` is the wb_comm.  Host buffers in, host buffers out; the collective itself runs on
// device staging buffers over xGMI (<= 16 KB per rank for an hour of audio: latency-bound, one hop on the mesh).
int wb_comm_allgather(void* user, const void* send, void* recv, int64_t bytes_per_rank) {
  wb_comm* c = static_cast<wb_comm*>(user);
  WB_REQUIRE(c && send && recv && bytes_per_rank > 0 && bytes_per_rank % 4 == 0, WB_ERR_ARG, "wb_comm_allgather: bad argument");
  Rccl* r;
  WB_TRY(rccl_load(&r));
  WB_HIP(hipSetDevice(c->device));
  WB_TRY(c->send.ensure((size_t)bytes_per_rank));
  WB_TRY(c->recv.ensure((size_t)bytes_per_rank * c->world));
  WB_HIP(hipMemcpyAsync(c->send.p, send, (size_t)bytes_per_rank, hipMemcpyHostToDevice, c->st));
  WB_NCCL(r, r->AllGather(c->send.p, c->recv.p, (size_t)bytes_per_rank / 4, NCCL_INT32, c->comm, c->st));
  WB_HIP(hipMemcpyAsync(recv, c->recv.p, (size_t)bytes_per_rank * c->world, hipMemcpyDeviceToHost, c->st));
  WB_HIP(hipStreamSynchronize(c->st));
  return WB_OK;
}

// Contiguous block of rank `rank`: [ceil(r K / R), ceil((r + 1) K / R))
int wb_shard_partition(int64_t n_windows, int rank, int world, int64_t* lo, int64_t* hi) {
  WB_REQUIRE(lo && hi && world >= 1 && rank >= 0 && rank < world && n_windows >= 0, WB_ERR_ARG, "wb_shard_partition: bad argument");
  *lo = (rank * n_windows + world - 1) / world;
  *hi = std::min<int64_t>(((rank + 1) * n_windows + world - 1) / world, n_windows);
  return WB_OK;
}

int wb_waveform_to_tokens_sharded(wb_model* m, const float* pcm, int pcm_on_device, int64_t n, int sample_rate,
                                  const wb_decode_params* p, const uint8_t* is_special, int rank, int world,
                                  wb_allgather_fn allgather, void* user, int32_t* win_tokens, int32_t row_stride,
                                  int32_t* win_lens, int64_t win_cap, int32_t* stitched, int64_t stitched_cap,
                                  int64_t* n_stitched) {
  WB_REQUIRE(m && pcm && p && is_special && win_tokens && win_lens && stitched && n_stitched, WB_ERR_ARG,
             "wb_waveform_to_tokens_sharded: null argument");
  WB_REQUIRE(world >= 1 && rank >= 0 && rank < world, WB_ERR_ARG, "rank %d outside [0, %d)", rank, world);
  WB_REQUIRE(world == 1 || allgather, WB_ERR_ARG, "wb_waveform_to_tokens_sharded: no all-gather for world size %d", world);
  WB_REQUIRE(row_stride >= 4 + p->max_depth, WB_ERR_ARG, "row_stride %d < %d", row_stride, 4 + p->max_depth);
  const int64_t wlen = wb_max_waveform_samples(m->max_mel_frames() - p->padding);   // transcribe.rs:32-34
  const int64_t K = wb_window_extents(n, sample_rate, wlen, p->overlap_seconds, nullptr, nullptr, 0);
  WB_REQUIRE(K <= win_cap, WB_ERR_ARG, "win_cap %lld < %lld windows", (long long)win_cap, (long long)K);
  int64_t lo, hi;
  WB_TRY(wb_shard_partition(K, rank, world, &lo, &hi));
  const int64_t rows = std::max<int64_t>(1, (K + world - 1) / world);               // rows every rank contributes
  const size_t rec = (size_t)1 + row_stride;                                         // [length | tokens]
  std::vector<int32_t> local((size_t)rows * rec, 0);
  for (int64_t i = 0; i < rows; i++) local[(size_t)i * rec] = -1;                    // -1: unused row
  int local_rc = WB_OK;
  std::string local_err;
  if (hi > lo) {
    std::vector<int32_t> toks((size_t)(hi - lo) * row_stride, 0), lens((size_t)(hi - lo), 0);
    const int rc = pcm_on_device
        ? wb_waveform_to_tokens_dev(m, pcm, n, sample_rate, p, is_special, (int)lo, (int)hi, toks.data(), row_stride, lens.data(),
                                    nullptr, 0, nullptr)
        : wb_waveform_to_tokens(m, pcm, n, sample_rate, p, is_special, (int)lo, (int)hi, toks.data(), row_stride, lens.data(),
                                nullptr, 0, nullptr);
    local_rc = rc;
    if (rc == WB_OK) {
      for (int64_t i = 0; i < hi - lo; i++) {
        local[(size_t)i * rec] = lens[(size_t)i];
        memcpy(&local[(size_t)i * rec + 1], &toks[(size_t)i * row_stride], (size_t)row_stride * 4);
      }
    } else {
      // A rank whose decode failed (out of memory, a HIP error) must still ENTER the collective: the other ranks are
      // already inside it (or about to be) and would block for ever.  It sends its rows with the sentinel length -2 and
      // its status in the first token slot; every rank then returns an error after the gather.
      local_err = wb_last_error();
      for (int64_t i = 0; i < rows; i++) { local[(size_t)i * rec] = -2; local[(size_t)i * rec + 1] = rc; }
    }
  }
  std::vector<int32_t> all((size_t)world * rows * rec);
  if (world == 1) all = local;
  else WB_TRY(allgather(user, local.data(), all.data(), (int64_t)(local.size() * 4)));   // the ONE exchange of the path
  WB_REQUIRE(local_rc == WB_OK, local_rc, "rank %d: local decode failed: %s", rank, local_err.c_str());
  for (int r = 0; r < world; r++) {
    const int32_t* row = &all[(size_t)r * rows * rec];
    WB_REQUIRE(row[0] != -2, WB_ERR_STATE, "rank %d reported a failed decode (status %d); no rank has a transcript", r, (int)row[1]);
  }
  // unpack in window order and fold the stitch over all K rows (transcribe.rs:56-63): identical on every rank
  int64_t w = 0;
  for (int r = 0; r < world; r++) {
    int64_t rlo, rhi;
    WB_TRY(wb_shard_partition(K, r, world, &rlo, &rhi));
    for (int64_t i = 0; i < rhi - rlo; i++, w++) {
      const int32_t* row = &all[((size_t)r * rows + (size_t)i) * rec];
      WB_REQUIRE(row[0] >= 0 && row[0] <= row_stride, WB_ERR_STATE, "rank %d sent no row for window %lld", r, (long long)(rlo + i));
      win_lens[w] = row[0];
      memcpy(win_tokens + (size_t)w * row_stride, row + 1, (size_t)row_stride * 4);
    }
  }
  WB_REQUIRE(w == K, WB_ERR_STATE, "gathered %lld of %lld windows", (long long)w, (long long)K);
  return wb_stitch_windows(win_tokens, row_stride, win_lens, K, p->max_n_offsets, p->min_n_overlaps, stitched, stitched_cap,
                           n_stitched);
}

}  // extern "C"
