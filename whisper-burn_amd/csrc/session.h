// Stateful decode session: mel -> encoder -> cross-K/V once per window batch, then KV-cached steps.
#pragma once
#include <vector>

#include "decode.h"
#include "engine.h"

struct wb_session {
  wb_model* m = nullptr;
  uint64_t model_uid = 0;       // m->uid at creation: what the pool is keyed by (m may dangle once its model is freed)
  int device = 0;               // m->device, kept so that a session can be destroyed after its model
  hipStream_t st = nullptr;
  hipStream_t st2 = nullptr;       // reads the chained decode's finished flags behind ev_seg while `st` runs ahead
  hipEvent_t ev_seg = nullptr;
  int W = 0, max_beams = 0, S = 0, padding = 0;
  double sample_rate = 16000.0;   // only feeds the mel filterbank (audio.rs:44)
  int Lmax = 0;                 // capacity (positions) of the self-KV cache / tables
  std::vector<int> T, C, row0;  // per window: mel frames (padded), encoder positions, first packed row
  std::vector<wb::MelWindow> wins_host; void* wins_dev_ptr = nullptr;   // what the device window table holds
  std::vector<int> meta_host; void* meta_dev_ptr = nullptr;             // ... and the window meta table
  std::vector<uint8_t> mask_host; void* mask_dev_ptr = nullptr;         // ... and the special-token mask (set_special_mask)
  int enc_rows = 0, maxC = 0, n_chunks = 0;
  wb::Workspace ws;
  wb::DevMem pcm, mel, wins, gmax, enc_out, ckv, win_meta;
  float* pcm_stage = nullptr; size_t pcm_stage_bytes = 0;   // pinned host staging of the caller's PCM (session_encode_pcm)
  wb::DevMem kc, vc, tabs, state;
  wb::StepLayout lay;
  char* host_block = nullptr;           // mapped pinned host memory: step state | top-k ids | top-k log-probs
  char* host_block_dev = nullptr;       // its device-visible address
  size_t host_bytes = 0;
  size_t chain_flags_off = 0;           // offset of the chained decode's per-row progress flags in host_block
  int* state_host = nullptr;            // views into host_block
  int32_t* topk_id_host = nullptr;      // [S][TOPK_MAX]
  float* topk_lp_host = nullptr;
  wb::DevMem x, h, att, Pqkv, Po, Pq, P1, P2, Pa, Pc, carec, ca, logits, tstats, row_stats, mask, lp_tmp, gctl, gtok, hm;
  int sk_qkv = 0, sk_o = 0, sk_1 = 0, sk_2 = 0;                             // its K-splits per weight shape (0: shape not served)
  wb::DevMem ps_layers, ps_roles, ps_ctl, ps_dead, ps_tstats, ps_stamps;   // persistent flag-chained decode (decode_persist.hip)
  wb::DevMem ps_gx, ps_gpa, ps_gpc, ps_gp2, ps_gxn;     // ... its residual streams / planes as 8-byte {tag, value} granules
  unsigned ps_launches = 0;                     // launches so far: the high half of the granule tags
  int ps_grid = -1;                                                    // co-resident blocks for this model (-1: not asked yet)
  int n_tiles_v = 0, ct_v = 128;
  int ks_qkv = 1, ksl_qkv = 0, ks_o = 1, ksl_o = 0, ks_1 = 1, ksl_1 = 0, ks_2 = 1, ksl_2 = 0, ks_v = 1, ksl_v = 0;
  wb::DevMem bc_ctl, bc_topk;   // device-chained beam search (decode.h: BeamChainArgs): control block, top-k rows of the step
  std::vector<int> prev_len, prev_win;
  int prev_n = 0, step = 0;
  bool has_mask = false, decode_ready = false;
  int last_use_mask = 0, last_had_logits = 0;
  // profiling only: which kernel classes carried the per-row cached K/V bytes of the last enqueued step, and how many
  // chained steps were enqueued since s->step was last advanced (the chained loop advances it once, at its end)
  int prof_cls_cross = -1, prof_cls_self = -1, prof_step_off = 0;
  // Range-guard words of the 16-bit matrix paths (mapped pinned host memory; the kernels raise them through guard_dev):
  // [0] the split-precision encoder-side GEMM of this session's encode pass (engine.cpp: split_guarded), [1] the
  // split-precision decoder GEMM of its batch-mode steps (dec_split_check).  Per session: sessions of one model on different
  // streams / threads never see or clear each other's flags.
  int* guard_host = nullptr; int* guard_dev = nullptr;
  bool enc_guard_pending = false;   // the encode pass was enqueued with a deferred check (session_enc_guard_resolve)
  wb::MelBatch enc_mb;              // ... its input, kept so the pass can be repeated on the exact-f32 kernel
  std::unordered_map<uint64_t, hipGraphExec_t> graphs;   // captured decode steps, keyed by launch shape
  uint64_t buf_sig = 0;                                  // signature of the buffers the graphs were captured with
  uint64_t beam_sig = 0;                                 // ... and of the constants of the last device-chained beam search
  void clear_graphs();
  ~wb_session();
};

namespace wb {
// profile accumulators (session.cpp)
struct Profile {
  bool on = false;
  double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
Profile& profile();
// per-kernel-class accumulators of the tagged launches (decode.h: prof_tag / WB_KLAUNCH)
struct KernelStat { int64_t calls = 0; double ms = 0, bytes = 0; };
void prof_adjust_bytes(int cls, double delta);   // correct a class's algorithmic bytes after the fact (dead rows)
void prof_collect();          // after the stream was synchronised: fold the pending tagged launches into the stats
// Timed region helper: records two events on `st` and adds the elapsed ms to slot `i` (when profiling is on).
struct ScopedTimer {
  hipStream_t st; int slot; hipEvent_t a = nullptr, b = nullptr; bool on;
  ScopedTimer(hipStream_t s, int slot_);
  void stop();          // records the end event
  void collect();       // after the stream was synchronised: accumulate
  ~ScopedTimer();
};

void session_pool_register(wb_model* m);   // a model became alive (build_model)
void session_pool_purge(wb_model* m);
int session_create(wb_model* m, int n_windows, int max_beams, int padding, wb_session** out);
// defer_guard: do not synchronise for the encoder's range guard; the caller calls session_enc_guard_resolve after its next
// synchronisation of s->st (the decode's) and, when told so, repeats the decode.
int session_encode_pcm(wb_session* s, const float* pcm, int64_t n_pcm, const int64_t* starts, const int64_t* lens,
                       bool pcm_on_device, bool defer_guard = false);
int session_enc_guard_resolve(wb_session* s, bool* reencoded);
void session_rewind(wb_session* s);      // back to step 0 over the same (re-)encoded window batch
int session_reserve(wb_session* s, int max_len);
// beam search with the bookkeeping on the device (decode.hip: dec_beam_update_kernel): *handled = false when the shape is not
// served (the caller runs the host-driven search)
int session_beam_chain(wb_session* s, const int32_t* prompt, int prompt_len, int beam_size, int eot, int max_depth,
                       int mask_until_len, int32_t* out_tokens, int32_t row_stride, int32_t* out_lens, bool* handled);
int session_greedy_chain(wb_session* s, const int32_t* prompt, int eot, int max_depth, int mask_until_len, int prompt_len,
                         int32_t* out_tokens, int32_t row_stride, int32_t* out_lens);
}  // namespace wb
