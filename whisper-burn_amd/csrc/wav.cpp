// WAV ingest with the reference's sample scaling.
//
// Replaces load_audio_waveform, /root/reference/src/bin/transcribe/main.rs:31-55 (hound reader):
//   * integer PCM sample s of b bits -> s as f32 / (2^(b-1) - 1) as f32   (main.rs:45, :49-52; NOT / 2^(b-1))
//   * IEEE float samples as they are                                        (main.rs:48)
//   * the sample rate must be 16 000 Hz and the file single-channel         (main.rs:42-43 asserts)
// hound semantics kept: 8-bit PCM is unsigned on disk and read as s - 128; 24-bit is packed little endian;
// WAVE_FORMAT_EXTENSIBLE resolves to its sub-format; chunks other than "fmt " / "data" are skipped
// (odd-sized chunks are padded to even).
#include <cmath>
#include <cstring>

#include "kernels.h"
#include "wb_internal.h"

using namespace wb;

namespace {

struct WavInfo {
  int64_t n_samples = 0;     // per channel
  int32_t sample_rate = 0, channels = 0, bits = 0, is_float = 0;
  long data_off = 0;
  int64_t data_bytes = 0;
};

uint32_t rd32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

int parse_header(FILE* f, const char* path, WavInfo* w) {
  unsigned char h[12];
  WB_REQUIRE(fread(h, 1, 12, f) == 12 && !memcmp(h, "RIFF", 4) && !memcmp(h + 8, "WAVE", 4), WB_ERR_IO,
             "%s: not a RIFF/WAVE file", path);
  bool have_fmt = false;
  for (;;) {
    unsigned char ch[8];
    WB_REQUIRE(fread(ch, 1, 8, f) == 8, WB_ERR_IO, "%s: no data chunk", path);
    const uint32_t size = rd32(ch + 4);
    if (!memcmp(ch, "fmt ", 4)) {
      unsigned char fm[40] = {0};
      const size_t want = size < sizeof(fm) ? size : sizeof(fm);
      WB_REQUIRE(size >= 16 && fread(fm, 1, want, f) == want, WB_ERR_IO, "%s: truncated fmt chunk", path);
      uint16_t tag = rd16(fm);
      w->channels = rd16(fm + 2);
      w->sample_rate = (int32_t)rd32(fm + 4);
      w->bits = rd16(fm + 14);
      if (tag == 0xFFFE) {   // WAVE_FORMAT_EXTENSIBLE: the first two bytes of the sub-format GUID are the real tag
        WB_REQUIRE(size >= 40, WB_ERR_IO, "%s: truncated extensible fmt chunk", path);
        tag = rd16(fm + 24);
      }
      WB_REQUIRE(tag == 1 || tag == 3, WB_ERR_IO, "%s: unsupported WAVE format tag %u (PCM and IEEE float only)", path, tag);
      w->is_float = tag == 3;
      WB_REQUIRE(w->is_float ? w->bits == 32 : (w->bits == 8 || w->bits == 16 || w->bits == 24 || w->bits == 32),
                 WB_ERR_IO, "%s: unsupported sample width %d", path, w->bits);
      have_fmt = true;
      if (size > want) fseek(f, (long)(size - want), SEEK_CUR);
      if (size & 1) fseek(f, 1, SEEK_CUR);
    } else if (!memcmp(ch, "data", 4)) {
      WB_REQUIRE(have_fmt && w->channels > 0, WB_ERR_IO, "%s: data chunk before fmt chunk", path);
      w->data_off = ftell(f);
      w->data_bytes = size;
      w->n_samples = (int64_t)size / (w->bits / 8) / w->channels;
      return WB_OK;
    } else {
      fseek(f, (long)(size + (size & 1)), SEEK_CUR);
    }
  }
}

// main.rs:45-52 for one sample of `bits` bits at p
inline float int_sample(const unsigned char* p, int bits, float max_int_val) {
  int32_t s;
  switch (bits) {
    case 8: s = (int32_t)p[0] - 128; break;                                        // hound: u8 on disk -> i8
    case 16: s = (int16_t)rd16(p); break;
    case 24: s = (int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24) >> 8; break;
    default: s = (int32_t)rd32(p); break;
  }
  return (float)s / max_int_val;
}

}  // namespace

extern "C" {

int wb_wav_info(const char* path, int64_t* n_samples, int32_t* sample_rate, int32_t* channels, int32_t* bits,
                int32_t* is_float) {
  WB_REQUIRE(path, WB_ERR_ARG, "wb_wav_info: null path");
  FILE* f = fopen(path, "rb");
  WB_REQUIRE(f, WB_ERR_IO, "cannot open %s", path);
  WavInfo w;
  const int rc = parse_header(f, path, &w);
  fclose(f);
  WB_TRY(rc);
  if (n_samples) *n_samples = w.n_samples;
  if (sample_rate) *sample_rate = w.sample_rate;
  if (channels) *channels = w.channels;
  if (bits) *bits = w.bits;
  if (is_float) *is_float = w.is_float;
  return WB_OK;
}

static int read_wav_f32(const char* path, float* out, int64_t capacity, int64_t* n_samples, bool require_16k);

int wb_wav_read_f32(const char* path, float* out, int64_t capacity, int64_t* n_samples) {
  return read_wav_f32(path, out, capacity, n_samples, true);
}

int wb_wav_read_f32_any_rate(const char* path, float* out, int64_t capacity, int64_t* n_samples) {
  return read_wav_f32(path, out, capacity, n_samples, false);
}

static int read_wav_f32(const char* path, float* out, int64_t capacity, int64_t* n_samples, bool require_16k) {
  WB_REQUIRE(path && out, WB_ERR_ARG, "wb_wav_read_f32: null argument");
  FILE* f = fopen(path, "rb");
  WB_REQUIRE(f, WB_ERR_IO, "cannot open %s", path);
  WavInfo w;
  int rc = parse_header(f, path, &w);
  if (rc != WB_OK) { fclose(f); return rc; }
  struct Closer { FILE* f; ~Closer() { fclose(f); } } closer{f};
  WB_REQUIRE(!require_16k || w.sample_rate == 16000, WB_ERR_SHAPE, "The audio sample rate must be 16k.");   // main.rs:42
  WB_REQUIRE(w.channels == 1, WB_ERR_SHAPE, "The audio must be single-channel.");            // main.rs:43
  WB_REQUIRE(w.n_samples <= capacity, WB_ERR_ARG, "wb_wav_read_f32: %lld samples > capacity %lld",
             (long long)w.n_samples, (long long)capacity);
  const int bps = w.bits / 8;
  const float max_int_val = (float)((1u << (w.bits - 1)) - 1u);                                 // main.rs:45
  std::vector<unsigned char> buf((size_t)1 << 20);
  int64_t done = 0;
  fseek(f, w.data_off, SEEK_SET);
  while (done < w.n_samples) {
    const size_t want = (size_t)std::min<int64_t>((int64_t)(buf.size() / bps), w.n_samples - done);
    WB_REQUIRE(fread(buf.data(), (size_t)bps, want, f) == want, WB_ERR_IO, "%s: truncated data chunk", path);
    if (w.is_float) {
      memcpy(out + done, buf.data(), want * 4);
    } else {
      for (size_t i = 0; i < want; i++) out[done + (int64_t)i] = int_sample(buf.data() + i * bps, w.bits, max_int_val);
    }
    done += (int64_t)want;
  }
  if (n_samples) *n_samples = w.n_samples;
  return WB_OK;
}

int wb_pcm_s16_to_f32_dev(int device, const int16_t* src_dev, int64_t n, float* dst_dev) {
  WB_REQUIRE(n >= 0 && (n == 0 || (src_dev && dst_dev)), WB_ERR_ARG, "wb_pcm_s16_to_f32_dev: bad argument");
  if (n == 0) return WB_OK;
  wb::GpuTurn turn(device);
  WB_HIP(hipSetDevice(device));
  launch_pcm_s16_to_f32(nullptr, src_dev, n, dst_dev);
  WB_HIP(hipGetLastError());
  WB_HIP(hipStreamSynchronize(nullptr));
  return WB_OK;
}

}  // extern "C"
