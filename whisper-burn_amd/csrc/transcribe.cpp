// Host restatement of the reference's L3 plumbing (src/transcribe.rs, src/beam.rs) over the
// session API: window extents, beam search, token-overlap stitch.  Pure integer / f64 host
// logic, restated exactly (property-tested against the Python restatement in oracle/).
#include <algorithm>
#include <cstring>
#include <string>

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

// ---- the reference's retired repetition detectors (transcribe.rs:385-447; dead code there, optional mode here) ----

int64_t wb_first_repetition_end(const int32_t* tokens, int64_t n, int64_t period) {
  // transcribe.rs:385-393; tokens.len() - period underflows (a panic there) -> WB_ERR_ARG
  WB_REQUIRE(tokens && period >= 0 && n >= period, WB_ERR_ARG, "wb_first_repetition_end: period %lld > %lld tokens",
             (long long)period, (long long)n);
  for (int64_t i = n - period - 1; i >= period; i--)
    if (memcmp(tokens + i - period, tokens + i, (size_t)period * sizeof(int32_t)) != 0) return i + 1;
  return period;
}

int64_t wb_repetition_period(const int32_t* tokens, int64_t n, int64_t min_repetitions) {
  // transcribe.rs:395-417; returns the period or 0 for None (a period is >= 1)
  if (!tokens || min_repetitions < 0) return 0;
  for (int64_t i = n - 1; i >= 0; i--) {
    const int64_t period = n - i;
    if (i / period < min_repetitions) return 0;
    bool all = true;
    for (int64_t j = 0; j < min_repetitions && all; j++) {
      const int64_t e = i - period * j, s = e - period;
      all = memcmp(tokens + s, tokens + i, (size_t)period * sizeof(int32_t)) == 0;
    }
    if (all) return period;
  }
  return 0;
}

int wb_find_repeated_tokens_index(const int32_t* tokens, int64_t n, int64_t window_size, int64_t min_repeat_count,
                                  int64_t* first_repeat_index, int64_t* end) {
  // transcribe.rs:419-447: the windows equal to the last one, the last window itself and anything overlapping it excluded
  WB_REQUIRE(tokens && window_size >= 1 && min_repeat_count >= 2, WB_ERR_ARG,
             "wb_find_repeated_tokens_index: window_size >= 1 and min_repeat_count >= 2 (the reference unwraps two matches)");
  if (2 * window_size > n) return 0;
  const int64_t last_index = n - window_size;
  int64_t n_repeats = 0, first = -1, second = -1;
  for (int64_t i = 0; i + window_size <= last_index; i++)
    if (memcmp(tokens + i, tokens + last_index, (size_t)window_size * sizeof(int32_t)) == 0) {
      if (n_repeats == 0) first = i;
      if (n_repeats == 1) second = i;
      n_repeats++;
    }
  if (n_repeats < min_repeat_count) return 0;
  if (first_repeat_index) *first_repeat_index = first;
  if (end) *end = second;
  return 1;
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

// ---- beam search over session steps (src/beam.rs:9-110 + transcribe.rs:253-312) -----------
namespace {

struct Beam {
  std::vector<int32_t> seq;
  double log_prob = 0.0;
  int prev_slot = -1;   // slot that held seq[0..len-1) in the last step it was fed
};

// beam.rs:81-110, literally (ascending list, insert before the first stored score >= score, evict index 0)
template <class T, class F>
std::vector<const T*> get_top_elements(const std::vector<T>& elems, F score, size_t num) {
  std::vector<const T*> top;
  std::vector<double> scores;
  for (const T& e : elems) {
    const double s = score(e);
    if (top.size() == num) {
      if (scores.empty() || s < scores[0]) continue;   // num == 0 would index out of bounds in the reference
    }
    size_t idx = scores.size();
    for (size_t i = 0; i < scores.size(); i++)
      if (scores[i] >= s) { idx = i; break; }
    top.insert(top.begin() + idx, &e);
    scores.insert(scores.begin() + idx, s);
    if (top.size() > num) { top.erase(top.begin()); scores.erase(scores.begin()); }
  }
  return top;
}

// Rust Iterator::max_by(partial_cmp().unwrap()): the LAST of equal maxima; NaN panics -> error
int max_by_log_prob(const std::vector<Beam>& beams, int* idx) {
  *idx = -1;
  for (size_t i = 0; i < beams.size(); i++) {
    WB_REQUIRE(beams[i].log_prob == beams[i].log_prob, WB_ERR_STATE, "beam search: NaN log-probability (reference panics)");
    if (*idx < 0 || beams[i].log_prob >= beams[*idx].log_prob) *idx = (int)i;
  }
  return WB_OK;
}

struct Cont { int32_t tok; double score; };

}  // namespace

// Beam search over an abstract step function (the session step on the GPU, or a caller's own).
// `prompt_in` / `P` (optional): the initial sequence of every window instead of the four-token prompt of
// transcribe.rs:203 -- the retired prompt-conditioning mode (transcribe.rs:188-199) prepends <|startofprev|> and
// the previous window's last non-special tokens.
static int beam_search_windows(const wb_decode_params* p, int W, int V, int S, wb_step_fn step, void* user,
                               int32_t* out_tokens, int32_t row_stride, int32_t* out_lens,
                               const int32_t* prompt_in = nullptr, int P = 4) {
  WB_REQUIRE(p && step && out_tokens && out_lens && W >= 1, WB_ERR_ARG, "beam search: null / bad argument");
  WB_REQUIRE(p->beam_size >= 1 && p->beam_size <= TOPK_MAX, WB_ERR_ARG, "beam_size %d outside [1, %d]", p->beam_size,
             TOPK_MAX);
  WB_REQUIRE(p->max_depth >= 0, WB_ERR_ARG, "max_depth must be >= 0");
  const int k = p->beam_size;
  const int32_t prompt4[4] = {p->tok_start_of_transcript, p->tok_language, p->tok_transcribe,
                              p->tok_no_timestamps};   // transcribe.rs:203
  const int32_t* prompt = prompt_in ? prompt_in : prompt4;
  if (!prompt_in) P = 4;
  WB_REQUIRE(P >= 1, WB_ERR_ARG, "beam search: empty prompt");
  for (int i = 0; i < P; i++) WB_REQUIRE(prompt[i] >= 0 && prompt[i] < V, WB_ERR_ARG, "prompt token %d out of range", prompt[i]);
  WB_REQUIRE(p->tok_end_of_text >= 0 && p->tok_end_of_text < V, WB_ERR_ARG, "end-of-text token out of range");
  WB_REQUIRE(row_stride >= P + p->max_depth, WB_ERR_ARG, "row_stride %d < %d", row_stride, P + p->max_depth);
  const int32_t eot = p->tok_end_of_text;
  auto finished = [&](const Beam& b) { return !b.seq.empty() && b.seq.back() == eot; };   // transcribe.rs:235-241

  // prefill: all prompt tokens but the last only feed the KV cache
  std::vector<int32_t> tok(W), par(W), win(W);
  for (int t = 0; t < P - 1; t++) {
    for (int w = 0; w < W; w++) { tok[w] = prompt[t]; par[w] = t == 0 ? -1 : w; win[w] = w; }
    WB_TRY(step(user, tok.data(), par.data(), win.data(), W, 0, 0, nullptr, nullptr));
  }
  std::vector<std::vector<Beam>> beams(W);
  std::vector<char> done(W, 0);
  for (int w = 0; w < W; w++) {
    Beam b;
    b.seq.assign(prompt, prompt + P); b.log_prob = 0.0; b.prev_slot = P > 1 ? w : -1;
    beams[w].push_back(std::move(b));
  }
  std::vector<int32_t> top_ids((size_t)S * k);
  std::vector<float> top_lp((size_t)S * k);
  for (int depth = 0; depth < p->max_depth; depth++) {   // beam.rs:22-31
    tok.clear(); par.clear(); win.clear();
    std::vector<std::vector<int>> slot_of(W);   // per window: slot of each beam (-1 for finished beams)
    for (int w = 0; w < W; w++) {
      if (done[w]) continue;
      int best;
      WB_TRY(max_by_log_prob(beams[w], &best));
      if (best >= 0 && finished(beams[w][best])) { done[w] = 1; continue; }   // beam.rs:23-27
      slot_of[w].assign(beams[w].size(), -1);
      for (size_t i = 0; i < beams[w].size(); i++) {
        if (finished(beams[w][i])) continue;   // the reference evaluates them too and discards the row (beam.rs:56-57)
        slot_of[w][i] = (int)tok.size();
        tok.push_back(beams[w][i].seq.back()); par.push_back(beams[w][i].prev_slot); win.push_back(w);
      }
    }
    if (tok.empty()) break;
    const int apply_mask = (P + depth) <= p->mask_until_len;   // transcribe.rs:271-275
    WB_TRY(step(user, tok.data(), par.data(), win.data(), (int)tok.size(), apply_mask, k, top_ids.data(),
                top_lp.data()));
    for (int w = 0; w < W; w++) {   // beam_search_step, beam.rs:39-79
      if (done[w]) continue;
      std::vector<Beam> finished_beams, new_beams;
      for (size_t i = 0; i < beams[w].size(); i++) {
        Beam& b = beams[w][i];
        if (finished(b)) { finished_beams.push_back(b); continue; }
        const int slot = slot_of[w][i];
        // the k best continuations by (log-prob desc, id asc) cover every element the V-wide
        // insertion scan (beam.rs:59, :81-110) can retain; replay that scan on them in id order
        std::vector<Cont> conts(k);
        for (int j = 0; j < k; j++)
          conts[j] = Cont{top_ids[(size_t)slot * k + j], b.log_prob + (double)top_lp[(size_t)slot * k + j]};   // :299
        std::sort(conts.begin(), conts.end(), [](const Cont& a, const Cont& c) { return a.tok < c.tok; });
        for (const Cont* c : get_top_elements(conts, [](const Cont& c) { return c.score; }, (size_t)k)) {
          Beam nb;
          nb.seq = b.seq; nb.seq.push_back(c->tok); nb.log_prob = c->score; nb.prev_slot = slot;
          new_beams.push_back(std::move(nb));
        }
      }
      std::vector<Beam> next;
      for (const Beam* b : get_top_elements(new_beams, [](const Beam& b) { return b.log_prob; }, (size_t)k)) next.push_back(*b);
      for (const Beam* b : get_top_elements(finished_beams, [](const Beam& b) { return b.log_prob; }, (size_t)k)) next.push_back(*b);
      beams[w] = std::move(next);
    }
  }
  for (int w = 0; w < W; w++) {   // beam.rs:33-36
    int best;
    WB_TRY(max_by_log_prob(beams[w], &best));
    const std::vector<int32_t> empty;
    const std::vector<int32_t>& seq = best >= 0 ? beams[w][best].seq : empty;
    WB_REQUIRE((int)seq.size() <= row_stride, WB_ERR_ARG, "row_stride too small");
    memcpy(out_tokens + (size_t)w * row_stride, seq.data(), seq.size() * sizeof(int32_t));
    out_lens[w] = (int32_t)seq.size();
  }
  return WB_OK;
}

static int session_step_thunk(void* user, const int32_t* new_tokens, const int32_t* parent, const int32_t* window,
                              int n, int apply_special_mask, int k, int32_t* top_ids, float* top_logprobs) {
  return wb_session_step(static_cast<wb_session*>(user), new_tokens, parent, window, n, apply_special_mask, k, top_ids,
                         top_logprobs);
}

extern "C" int wb_session_decode(wb_session* s, const wb_decode_params* p, int32_t* out_tokens, int32_t row_stride,
                                 int32_t* out_lens) {
  WB_REQUIRE(s && p && out_tokens && out_lens, WB_ERR_ARG, "wb_session_decode: null argument");
  wb::GpuTurn turn(s->device);
  WB_REQUIRE(p->beam_size >= 1 && p->beam_size <= s->max_beams, WB_ERR_ARG, "beam_size %d outside [1, %d]",
             p->beam_size, s->max_beams);
  if (!s->decode_ready || s->Lmax < 4 + p->max_depth) WB_TRY(session_reserve(s, 4 + p->max_depth + 1));
  static const bool chain_enabled = []() { const char* e = getenv("WHISPER_HIP_CHAIN"); return !(e && e[0] == '0'); }();
  if (p->beam_size == 1 && s->max_beams == 1 && chain_enabled && p->max_depth > 0 && s->step == 0) {
    // greedy: prompt prefill through the ordinary step, then the device-chained loop (no per-step host round trip)
    const int V = s->m->dims.n_vocab, W = s->W;
    const int32_t prompt[4] = {p->tok_start_of_transcript, p->tok_language, p->tok_transcribe, p->tok_no_timestamps};
    for (int t : prompt) WB_REQUIRE(t >= 0 && t < V, WB_ERR_ARG, "prompt token %d out of range", t);
    WB_REQUIRE(p->tok_end_of_text >= 0 && p->tok_end_of_text < V, WB_ERR_ARG, "end-of-text token out of range");
    WB_REQUIRE(row_stride >= 4 + p->max_depth, WB_ERR_ARG, "row_stride %d < %d", row_stride, 4 + p->max_depth);
    for (int w = 0; w < W; w++) memcpy(out_tokens + (size_t)w * row_stride, prompt, sizeof(prompt));
    // (the prompt prefill happens inside: as forced steps of the persistent launch, or as host-driven steps in front of the chain)
    return session_greedy_chain(s, prompt, p->tok_end_of_text, p->max_depth, p->mask_until_len, 4, out_tokens,
                                row_stride, out_lens);
  }
  if (s->step == 0) {
    // beam search with the bookkeeping on the device (session.cpp: session_beam_chain; WHISPER_HIP_BEAM_CHAIN=0: host-driven)
    const int V = s->m->dims.n_vocab;
    const int32_t prompt[4] = {p->tok_start_of_transcript, p->tok_language, p->tok_transcribe, p->tok_no_timestamps};
    for (int t : prompt) WB_REQUIRE(t >= 0 && t < V, WB_ERR_ARG, "prompt token %d out of range", t);
    WB_REQUIRE(p->tok_end_of_text >= 0 && p->tok_end_of_text < V, WB_ERR_ARG, "end-of-text token out of range");
    WB_REQUIRE(row_stride >= 4 + p->max_depth, WB_ERR_ARG, "row_stride %d < %d", row_stride, 4 + p->max_depth);
    bool handled = false;
    WB_TRY(session_beam_chain(s, prompt, 4, p->beam_size, p->tok_end_of_text, p->max_depth, p->mask_until_len, out_tokens,
                              row_stride, out_lens, &handled));
    if (handled) return WB_OK;
  }
  return beam_search_windows(p, s->W, s->m->dims.n_vocab, s->S, session_step_thunk, s, out_tokens, row_stride, out_lens);
}

extern "C" int wb_session_decode_prompt(wb_session* s, const wb_decode_params* p, const int32_t* prompt,
                                        int32_t prompt_len, int32_t* out_tokens, int32_t row_stride, int32_t* out_lens) {
  WB_REQUIRE(s && p && prompt && out_tokens && out_lens, WB_ERR_ARG, "wb_session_decode_prompt: null argument");
  WB_REQUIRE(prompt_len >= 1, WB_ERR_ARG, "wb_session_decode_prompt: empty prompt");
  WB_REQUIRE(p->beam_size >= 1 && p->beam_size <= s->max_beams, WB_ERR_ARG, "beam_size %d outside [1, %d]",
             p->beam_size, s->max_beams);
  if (!s->decode_ready || s->Lmax < prompt_len + p->max_depth) WB_TRY(session_reserve(s, prompt_len + p->max_depth + 1));
  return beam_search_windows(p, s->W, s->m->dims.n_vocab, s->S, session_step_thunk, s, out_tokens, row_stride, out_lens,
                             prompt, prompt_len);
}

// transcribe.rs:23-74 with the prompt conditioning of :43-50 / :188-199 switched back on (the reference shadows it
// with an empty list at :201 because "including the prev tokens causes whisper to hallucinate", :186).  Windows
// depend on their predecessor's tokens here, so they are decoded one after another, one session each.
extern "C" int wb_waveform_to_tokens_prompted(wb_model* m, const float* pcm, int64_t n, int sample_rate,
                                              const wb_decode_params* p, const uint8_t* is_special,
                                              int32_t tok_start_of_prev, int32_t n_prev_tokens, int32_t* win_tokens,
                                              int32_t row_stride, int32_t* win_lens, int32_t* stitched,
                                              int64_t stitched_cap, int64_t* n_stitched) {
  WB_REQUIRE(m && pcm && p && is_special && win_tokens && win_lens && stitched && n_stitched, WB_ERR_ARG,
             "wb_waveform_to_tokens_prompted: null argument");
  const int V = m->dims.n_vocab;
  WB_REQUIRE(tok_start_of_prev >= 0 && tok_start_of_prev < V, WB_ERR_ARG, "start-of-prev token out of range");
  WB_REQUIRE(n_prev_tokens >= 0 && n_prev_tokens <= 64, WB_ERR_ARG, "n_prev_tokens outside [0, 64]");
  WB_REQUIRE(p->padding >= 0 && p->padding < m->max_mel_frames(), WB_ERR_ARG, "bad padding");
  const int max_prompt = 1 + n_prev_tokens + 4;
  WB_REQUIRE(row_stride >= max_prompt + p->max_depth, WB_ERR_ARG, "row_stride %d < %d", row_stride,
             max_prompt + p->max_depth);
  const int64_t wlen = wb_max_waveform_samples(m->max_mel_frames() - p->padding);   // transcribe.rs:32-34
  const int64_t n_win = wb_window_extents(n, sample_rate, wlen, p->overlap_seconds, nullptr, nullptr, 0);
  std::vector<int64_t> starts((size_t)n_win), lens((size_t)n_win);
  wb_window_extents(n, sample_rate, wlen, p->overlap_seconds, starts.data(), lens.data(), n_win);
  std::vector<int32_t> tokens;   // transcribe.rs:40
  for (int64_t w = 0; w < n_win; w++) {
    // transcribe.rs:43-50: the last n non-special tokens so far, oldest first
    std::vector<int32_t> prev;
    for (auto it = tokens.rbegin(); it != tokens.rend() && (int)prev.size() < n_prev_tokens; ++it)
      if (!is_special[*it]) prev.push_back(*it);
    std::reverse(prev.begin(), prev.end());
    // transcribe.rs:188-199, :203
    std::vector<int32_t> prompt;
    if (!prev.empty()) { prompt.push_back(tok_start_of_prev); prompt.insert(prompt.end(), prev.begin(), prev.end()); }
    for (int32_t t : {p->tok_start_of_transcript, p->tok_language, p->tok_transcribe, p->tok_no_timestamps}) prompt.push_back(t);
    wb_session* s = nullptr;
    int rc = session_create(m, 1, p->beam_size, p->padding, &s);
    int32_t* row = win_tokens + (size_t)w * row_stride;
    if (rc == WB_OK) {
      s->sample_rate = (double)sample_rate;
      rc = session_encode_pcm(s, pcm, n, &starts[(size_t)w], &lens[(size_t)w], false);
      if (rc == WB_OK) rc = wb_session_set_special_mask(s, is_special);
      if (rc == WB_OK) rc = wb_session_decode_prompt(s, p, prompt.data(), (int32_t)prompt.size(), row, row_stride, &win_lens[w]);
      wb_session_free(s);
    }
    WB_TRY(rc);
    // transcribe.rs:56-63
    int64_t pi = 0, ci = 0;
    if (wb_find_chunk_overlap(tokens.data(), (int64_t)tokens.size(), row, win_lens[w], p->max_n_offsets, p->min_n_overlaps,
                              &pi, &ci)) {
      tokens.resize((size_t)pi);
      tokens.insert(tokens.end(), row + ci, row + win_lens[w]);
    } else {
      tokens.insert(tokens.end(), row, row + win_lens[w]);
    }
  }
  WB_REQUIRE((int64_t)tokens.size() <= stitched_cap, WB_ERR_ARG, "stitched capacity %lld < %zu", (long long)stitched_cap,
             tokens.size());
  memcpy(stitched, tokens.data(), tokens.size() * sizeof(int32_t));
  *n_stitched = (int64_t)tokens.size();
  return WB_OK;
}

extern "C" int wb_beam_search(const wb_decode_params* p, int n_windows, int n_vocab, wb_step_fn step, void* user,
                              int32_t* out_tokens, int32_t row_stride, int32_t* out_lens) {
  WB_REQUIRE(p, WB_ERR_ARG, "wb_beam_search: null params");
  return beam_search_windows(p, n_windows, n_vocab, n_windows * std::max(1, p->beam_size), step, user, out_tokens,
                             row_stride, out_lens);
}

static int waveform_to_tokens_impl(wb_model* m, const float* pcm, bool pcm_on_device, int64_t n, int sample_rate,
                                   const wb_decode_params* p, const uint8_t* is_special, int win_begin, int win_end,
                                   int32_t* win_tokens, int32_t row_stride, int32_t* win_lens, int32_t* stitched,
                                   int64_t stitched_cap, int64_t* n_stitched) {
  WB_REQUIRE(m && pcm && p && is_special && win_tokens && win_lens, WB_ERR_ARG, "wb_waveform_to_tokens: null argument");
  wb::GpuTurn turn(m->device);   // (the sharded entry point calls this for its local windows and exchanges results outside the turn)
  WB_REQUIRE(p->padding >= 0 && p->padding < m->max_mel_frames(), WB_ERR_ARG, "bad padding");
  // transcribe.rs:32-34
  const int64_t wlen = wb_max_waveform_samples(m->max_mel_frames() - p->padding);
  const int64_t n_win = wb_window_extents(n, sample_rate, wlen, p->overlap_seconds, nullptr, nullptr, 0);
  std::vector<int64_t> starts((size_t)n_win), lens((size_t)n_win);
  wb_window_extents(n, sample_rate, wlen, p->overlap_seconds, starts.data(), lens.data(), n_win);
  if (win_end < 0 || win_end > n_win) win_end = (int)n_win;
  win_begin = std::max(0, std::min(win_begin, win_end));
  const int n_local = win_end - win_begin;
  // windows are independent (transcribe.rs:195-201): they are decoded in batches of up to 64, one session each
  // (several sessions on concurrent host threads were measured: +9 % at two, slower beyond -- not kept)
  const int batch = p->max_batch_windows > 0 ? p->max_batch_windows : 64;
  const int n_batches = (n_local + batch - 1) / batch;
  auto run_batch = [&](int bi) -> int {
    const int b0 = bi * batch, nb = std::min(batch, n_local - b0);
    wb_session* s = nullptr;
    int rc = session_create(m, nb, p->beam_size, p->padding, &s);
    if (rc == WB_OK) {
      s->sample_rate = (double)sample_rate;
      // the encoder's range guard is resolved behind the decode's own synchronisation (no extra one per batch); if the
      // split-precision kernel tripped, the pass has been repeated on the exact-f32 kernel and the batch is decoded again
      rc = session_encode_pcm(s, pcm, n, starts.data() + win_begin + b0, lens.data() + win_begin + b0, pcm_on_device, true);
      if (rc == WB_OK) rc = wb_session_set_special_mask(s, is_special);
      if (rc == WB_OK) rc = session_reserve(s, 4 + p->max_depth + 1);
      for (int attempt = 0; rc == WB_OK && attempt < 2; attempt++) {
        rc = wb_session_decode(s, p, win_tokens + (size_t)b0 * row_stride, row_stride, win_lens + b0);
        bool reencoded = false;
        const int rg = session_enc_guard_resolve(s, &reencoded);
        if (rg != WB_OK) { rc = rg; break; }
        if (!reencoded) break;
        rc = WB_OK;                            // (whatever the decode made of the non-finite encoder output is void)
        session_rewind(s);                     // back to step 0 over the re-encoded window batch
      }
      wb_session_free(s);
    }
    return rc;
  };
  for (int bi = 0; bi < n_batches; bi++) WB_TRY(run_batch(bi));
  if (stitched) {
    WB_REQUIRE(n_stitched, WB_ERR_ARG, "n_stitched is null");
    WB_TRY(wb_stitch_windows(win_tokens, row_stride, win_lens, n_local, p->max_n_offsets, p->min_n_overlaps, stitched,
                             stitched_cap, n_stitched));
  }
  return WB_OK;
}

extern "C" int wb_waveform_to_tokens(wb_model* m, const float* pcm, int64_t n, int sample_rate,
                                     const wb_decode_params* p, const uint8_t* is_special, int win_begin, int win_end,
                                     int32_t* win_tokens, int32_t row_stride, int32_t* win_lens, int32_t* stitched,
                                     int64_t stitched_cap, int64_t* n_stitched) {
  return waveform_to_tokens_impl(m, pcm, false, n, sample_rate, p, is_special, win_begin, win_end, win_tokens, row_stride,
                                 win_lens, stitched, stitched_cap, n_stitched);
}

extern "C" int wb_waveform_to_tokens_dev(wb_model* m, const float* pcm_dev, int64_t n, int sample_rate,
                                         const wb_decode_params* p, const uint8_t* is_special, int win_begin,
                                         int win_end, int32_t* win_tokens, int32_t row_stride, int32_t* win_lens,
                                         int32_t* stitched, int64_t stitched_cap, int64_t* n_stitched) {
  return waveform_to_tokens_impl(m, pcm_dev, true, n, sample_rate, p, is_special, win_begin, win_end, win_tokens,
                                 row_stride, win_lens, stitched, stitched_cap, n_stitched);
}
