// Encoder / decoder orchestration over the gfx950 kernels (host side).
#pragma once
#include <functional>
#include <vector>

#include "kernels.h"
#include "wb_internal.h"

namespace wb {

// Geometry of a batch of mel windows laid out [n_windows][80][row_stride]; T[w] = frames incl. padding.
struct MelBatch {
  const float* mel = nullptr;      // device
  int64_t win_stride = 0;          // floats between windows
  int row_stride = 0;              // floats between mel rows
  std::vector<int> T;              // frames per window (<= n_audio_ctx)
};

struct EncoderOut {
  std::vector<int> C;              // encoder positions per window: (T-1)/2+1  (mod.rs:244 stride-2 conv)
  std::vector<int> row0;           // first packed row of each window
  int rows = 0;                    // sum C
};

// AudioEncoder::forward (mod.rs:228-260) for a packed batch of ragged windows -> out[rows][d].
// (guarded: see split_guarded)
int run_encoder(wb_model* m, hipStream_t st, Workspace& ws, const MelBatch& mb, float* out_dev, EncoderOut* eo);
int run_encoder_unguarded(wb_model* m, hipStream_t st, Workspace& ws, const MelBatch& mb, float* out_dev, EncoderOut* eo);

// Run `body` (a pass that may use the split-precision GEMM, gemm_f16x3.hip) under the range-flag word `flag_host` /
// `flag_dev` (one mapped host word per pass owner: session or model); if the kernel raised it -- an activation outside
// fp16's range -- the model switches to the exact-f32 kernel for good and the pass runs again.  `deferred` non-null: no
// synchronisation here; the caller resolves the flag at its next one (split_guard_resolve) and repeats the work itself.
int split_guarded(wb_model* m, hipStream_t st, int* flag_host, int* flag_dev, bool* deferred,
                  const std::function<int()>& body);
bool split_guard_resolve(wb_model* m, int* flag_host);

// TextDecoder::forward (mod.rs:131-157), stateless: tokens_dev [n*L], enc_dev [n*C][d] -> logits_dev [n*L][V].
int run_decoder_stateless(wb_model* m, hipStream_t st, Workspace& ws, const int32_t* tokens_dev, int n, int L,
                          const float* enc_dev, int C, float* logits_dev);

// split-precision fp16 MFMA GEMM when split copies sh / sl ([N][ldwt] fp16) are given and the shape fits, else exact-f32 MFMA
int gemm_dispatch(const wb_model* m, hipStream_t st, const GemmArgs& a, int ldwt, const uint16_t* sh = nullptr,
                  const uint16_t* sl = nullptr);

// Process-wide mel constant tables for (device, sample_rate).
int get_mel_tables(int device, double sample_rate, const MelTables** out_dev);
// developer tool (WHISPER_HIP_ENC_TRACE): a stream-ordered copy of one more stage into this thread's encoder trace
void enc_trace_stage(hipStream_t st, const char* name, const void* p, size_t bytes);

}  // namespace wb
