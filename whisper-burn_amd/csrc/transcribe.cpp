// Host restatement of the reference's L3 plumbing (src/transcribe.rs, src/beam.rs) over the
// session API: window extents, beam search, token-overlap stitch.  Pure integer / f64 host
// logic, restated exactly (property-tested against the Python restatement in oracle/).
#include <algorithm>
#include <cstring>

#include "engine.h"
#include "session.h"

using namespace wb;

extern "C" {

void wb_decode_params_default(wb_decode_params* p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->beam_size = 5;          // transcribe.rs:232
  p->max_depth = 100;        // transcribe.rs:233
  p->padding = 10;           // transcribe.rs:33
  p->overlap_seconds = 3;    // transcribe.rs:120
  p->max_n_offsets = 40;     // transcribe.rs:57
  p->min_n_overlaps = 3;     // transcribe.rs:57
  p->mask_until_len = 5;     // transcribe.rs:271
  p->max_batch_windows = 0;
  p->tok_start_of_transcript = p->tok_language = p->tok_transcribe = p->tok_no_timestamps =
      p->tok_end_of_text = -1;
}

int64_t wb_window_extents(int64_t n_samples, int sample_rate, int64_t window_len, int overlap_seconds,
                          int64_t* starts, int64_t* lens, int64_t cap) {
  // transcribe.rs:120-128
  const int64_t chunk_overlap = (int64_t)sample_rate * overlap_seconds;
  const int64_t shift = std::max<int64_t>(std::max<int64_t>(window_len - chunk_overlap, 0), 1);
  const int64_t iter_len = std::max<int64_t>(n_samples - 1, 0) / shift + 1;
  if (starts && lens) {
    for (int64_t i = 0; i < iter_len && i < cap; i++) {
      const int64_t start = i * shift;
      const int64_t end = std::min(start + window_len, n_samples);
      starts[i] = start;
      lens[i] = end - start;
    }
  }
  return iter_len;
}

int wb_find_chunk_overlap(const int32_t* prev, int64_t n_prev, const int32_t* curr, int64_t n_curr,
                          int max_n_offsets, int min_n_overlaps, int64_t* prev_index, int64_t* curr_index) {
  // transcribe.rs:76-110
  int64_t max_overlap = 0, best_prev = 0, best_curr = 0;
  const int64_t n_offsets = std::min<int64_t>(std::min(n_prev, n_curr), max_n_offsets);
  for (int64_t offset = 0; offset < n_offsets; offset++) {
    const int64_t prev_start = n_prev - 1 - offset;
    const int64_t span = std::min(n_prev - prev_start, n_curr);   // zip stops at the shorter side
    int64_t n_overlap = 0, first = -1;
    for (int64_t i = 0; i < span; i++)
      if (prev[prev_start + i] == curr[i]) {
        if (first < 0) first = i;
        n_overlap++;
      }
    if (n_overlap > max_overlap) {
      max_overlap = n_overlap;
      best_curr = first;
      best_prev = prev_start + first;
    }
  }
  if (max_overlap >= min_n_overlaps) {
    if (prev_index) *prev_index = best_prev;
    if (curr_index) *curr_index = best_curr;
    return 1;
  }
  return 0;
}

int wb_stitch_windows(const int32_t* win_tokens, int32_t row_stride, const int32_t* win_lens, int n_windows,
                      int max_n_offsets, int min_n_overlaps, int32_t* out, int64_t cap, int64_t* n_out) {
  WB_REQUIRE(win_lens && out && n_out && (win_tokens || n_windows == 0), WB_ERR_ARG, "wb_stitch_windows: null argument");
  std::vector<int32_t> tokens;
  for (int w = 0; w < n_windows; w++) {
    const int32_t* nt = win_tokens + (int64_t)w * row_stride;
    const int64_t nn = win_lens[w];
    int64_t pi, ci;
    // transcribe.rs:56-63
    if (wb_find_chunk_overlap(tokens.data(), (int64_t)tokens.size(), nt, nn, max_n_offsets, min_n_overlaps, &pi,
                              &ci)) {
      tokens.resize((size_t)pi);
      tokens.insert(tokens.end(), nt + ci, nt + nn);
    } else {
      tokens.insert(tokens.end(), nt, nt + nn);
    }
  }
  WB_REQUIRE((int64_t)tokens.size() <= cap, WB_ERR_ARG, "wb_stitch_windows: output capacity %lld < %zu",
             (long long)cap, tokens.size());
  memcpy(out, tokens.data(), tokens.size() * sizeof(int32_t));
  *n_out = (int64_t)tokens.size();
  return WB_OK;
}

}  // extern "C"
