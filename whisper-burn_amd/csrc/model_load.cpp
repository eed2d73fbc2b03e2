// Weight ingestion: the reference's dump-directory format -> device arena.
//
// Format (written by /root/reference/python/dump.py:130-210, read by
// /root/reference/src/model/load.rs:19-310): every tensor is a 1-D little-endian f32 .npy
// whose contents are [dim_0 .. dim_{D-1}, v_0, v_1, ...]; scalars are [1.0, value];
// Linear weights are stored [d_in, d_out] (dump.py:141-145); attention keys have no bias
// file (dump.py:144, load.rs:57 `.ok()`).  Tensor names = path relative to the dump dir.
#include <cmath>
#include <cstring>
#include <filesystem>
#include <fstream>

#include "wb_internal.h"
#include "kernels.h"
#include "decode.h"

namespace wb { void session_pool_register(wb_model* m); }

namespace fs = std::filesystem;

namespace wb {

static int parse_npy(const std::string& path, HostTensor* t) {
  std::ifstream f(path, std::ios::binary);
  WB_REQUIRE(f.good(), WB_ERR_IO, "cannot open %s", path.c_str());
  f.seekg(0, std::ios::end);
  size_t sz = (size_t)f.tellg();
  f.seekg(0);
  std::vector<char> buf(sz);
  f.read(buf.data(), (std::streamsize)sz);
  WB_REQUIRE(sz >= 10 && memcmp(buf.data(), "\x93NUMPY", 6) == 0, WB_ERR_IO, "%s: not an .npy file",
             path.c_str());
  int major = (unsigned char)buf[6];
  size_t hlen, hoff;
  if (major == 1) {
    hlen = (unsigned char)buf[8] | ((unsigned char)buf[9] << 8);
    hoff = 10;
  } else {
    WB_REQUIRE(sz >= 12, WB_ERR_IO, "%s: truncated header", path.c_str());
    hlen = (unsigned char)buf[8] | ((unsigned char)buf[9] << 8) | ((unsigned char)buf[10] << 16) |
           ((size_t)(unsigned char)buf[11] << 24);
    hoff = 12;
  }
  WB_REQUIRE(hoff + hlen <= sz, WB_ERR_IO, "%s: truncated header", path.c_str());
  std::string hdr(buf.data() + hoff, hlen);
  WB_REQUIRE(hdr.find("'<f4'") != std::string::npos || hdr.find("'f4'") != std::string::npos,
             WB_ERR_IO, "%s: dtype is not little-endian float32", path.c_str());
  WB_REQUIRE(hdr.find("'fortran_order': False") != std::string::npos, WB_ERR_IO,
             "%s: fortran_order must be False", path.c_str());
  size_t sp = hdr.find("'shape':");
  WB_REQUIRE(sp != std::string::npos, WB_ERR_IO, "%s: no shape in header", path.c_str());
  size_t lp = hdr.find('(', sp), rp = hdr.find(')', sp);
  WB_REQUIRE(lp != std::string::npos && rp != std::string::npos, WB_ERR_IO, "%s: bad shape",
             path.c_str());
  int64_t count = 1;
  {
    std::string s = hdr.substr(lp + 1, rp - lp - 1);
    size_t pos = 0;
    bool any = false;
    while (pos < s.size()) {
      while (pos < s.size() && (s[pos] == ' ' || s[pos] == ',')) pos++;
      if (pos >= s.size()) break;
      count *= std::strtoll(s.c_str() + pos, nullptr, 10);
      any = true;
      while (pos < s.size() && s[pos] != ',') pos++;
    }
    if (!any) count = 1;
  }
  size_t doff = hoff + hlen;
  WB_REQUIRE(doff + (size_t)count * 4 <= sz, WB_ERR_IO, "%s: truncated data", path.c_str());
  t->owned.resize((size_t)count);
  memcpy(t->owned.data(), buf.data() + doff, (size_t)count * 4);
  t->shape = {count};   // still flat: dims-prefix decoded later, per expected rank (load.rs:19-27)
  t->data = t->owned.data();
  return WB_OK;
}

int read_dump_dir(const char* dir, TensorMap* out) {
  std::error_code ec;
  WB_REQUIRE(fs::is_directory(dir, ec), WB_ERR_IO, "%s is not a directory", dir);
  fs::path root(dir);
  for (auto it = fs::recursive_directory_iterator(root, ec); !ec && it != fs::recursive_directory_iterator();
       it.increment(ec)) {
    if (!it->is_regular_file()) continue;
    fs::path p = it->path();
    if (p.extension() != ".npy") continue;
    std::string name = fs::relative(p, root).generic_string();
    name.resize(name.size() - 4);
    HostTensor t;
    WB_TRY(parse_npy(p.string(), &t));
    // flat, dims-prefixed: mark with an empty shape so the builder decodes the prefix
    t.shape.clear();
    (*out)[name] = std::move(t);
    (*out)[name].data = (*out)[name].owned.data();
  }
  WB_REQUIRE(!ec, WB_ERR_IO, "walking %s: %s", dir, ec.message().c_str());
  return WB_OK;
}

// ---- builder --------------------------------------------------------------------
struct Builder {
  TensorMap& tm;
  std::vector<float> host;   // staging image of the whole arena
  explicit Builder(TensorMap& t) : tm(t) {}

  // Fetch a tensor with the rank the reference loader expects (load.rs load_tensor::<B, D>).
  int get(const std::string& name, int rank, const float** data, std::vector<int64_t>* shape,
          bool optional = false) {
    auto it = tm.find(name);
    if (it == tm.end()) {
      if (optional) { *data = nullptr; return WB_OK; }
      set_error("missing tensor %s", name.c_str());
      return WB_ERR_IO;
    }
    HostTensor& t = it->second;
    if (t.shape.empty()) {   // dims-prefixed flat dump
      int64_t n = (int64_t)t.owned.size();
      WB_REQUIRE(n >= rank, WB_ERR_IO, "%s: shorter than its rank", name.c_str());
      shape->clear();
      int64_t numel = 1;
      for (int i = 0; i < rank; i++) {
        int64_t d = (int64_t)t.data[i];
        shape->push_back(d);
        numel *= d;
      }
      WB_REQUIRE(numel == n - rank, WB_ERR_IO, "%s: dims prefix (%d) does not match payload", name.c_str(),
                 rank);
      *data = t.data + rank;
    } else {
      WB_REQUIRE((int)t.shape.size() == rank, WB_ERR_SHAPE, "%s: expected rank %d, got %zu", name.c_str(),
                 rank, t.shape.size());
      *shape = t.shape;
      *data = t.data;
    }
    return WB_OK;
  }
  int scalar(const std::string& name, float* v) {
    const float* d; std::vector<int64_t> s;
    WB_TRY(get(name, 1, &d, &s));
    WB_REQUIRE(s[0] >= 1, WB_ERR_IO, "%s: empty scalar", name.c_str());
    *v = d[0];
    return WB_OK;
  }
  // reserve `n` floats in the arena image (64-float aligned = 256 B), return offset
  size_t reserve(size_t n) {
    size_t off = (host.size() + 63) & ~size_t(63);
    host.resize(off + n, 0.f);
    return off;
  }
};

struct Off { size_t w = SIZE_MAX, b = SIZE_MAX; int k = 0, n = 0; };

static int put_linear(Builder& B, const std::string& p, int k, int n, Off* o, bool bias_required) {
  const float* w; std::vector<int64_t> s;
  WB_TRY(B.get(p + "/weight", 2, &w, &s));
  WB_REQUIRE(s[0] == k && s[1] == n, WB_ERR_SHAPE, "%s/weight: expected [%d,%d], got [%lld,%lld]", p.c_str(),
             k, n, (long long)s[0], (long long)s[1]);
  o->k = k; o->n = n;
  o->w = B.reserve((size_t)k * n);
  memcpy(&B.host[o->w], w, sizeof(float) * (size_t)k * n);
  const float* b; std::vector<int64_t> sb;
  WB_TRY(B.get(p + "/bias", 1, &b, &sb, !bias_required));
  o->b = B.reserve((size_t)n);
  if (b) {
    WB_REQUIRE(sb[0] == n, WB_ERR_SHAPE, "%s/bias: expected [%d]", p.c_str(), n);
    memcpy(&B.host[o->b], b, sizeof(float) * (size_t)n);
  }
  return WB_OK;
}

struct LnOff { size_t g, b; float eps; };
static int put_ln(Builder& B, const std::string& p, int n, LnOff* o) {
  const float *g, *b; std::vector<int64_t> s;
  WB_TRY(B.get(p + "/weight", 1, &g, &s));
  WB_REQUIRE(s[0] == n, WB_ERR_SHAPE, "%s/weight: expected [%d]", p.c_str(), n);
  WB_TRY(B.get(p + "/bias", 1, &b, &s));
  WB_REQUIRE(s[0] == n, WB_ERR_SHAPE, "%s/bias: expected [%d]", p.c_str(), n);
  WB_TRY(B.scalar(p + "/eps", &o->eps));
  o->g = B.reserve(n); memcpy(&B.host[o->g], g, sizeof(float) * n);
  o->b = B.reserve(n); memcpy(&B.host[o->b], b, sizeof(float) * n);
  return WB_OK;
}

// q|k|v -> one [d][3d] matrix (+ bias [3d], key part zero)
static int put_qkv(Builder& B, const std::string& p, int d, Off* o) {
  const float *q, *k, *v, *bq, *bv; std::vector<int64_t> s;
  WB_TRY(B.get(p + "/query/weight", 2, &q, &s));
  WB_REQUIRE(s[0] == d && s[1] == d, WB_ERR_SHAPE, "%s/query/weight: expected [%d,%d]", p.c_str(), d, d);
  WB_TRY(B.get(p + "/key/weight", 2, &k, &s));
  WB_REQUIRE(s[0] == d && s[1] == d, WB_ERR_SHAPE, "%s/key/weight: expected [%d,%d]", p.c_str(), d, d);
  WB_TRY(B.get(p + "/value/weight", 2, &v, &s));
  WB_REQUIRE(s[0] == d && s[1] == d, WB_ERR_SHAPE, "%s/value/weight: expected [%d,%d]", p.c_str(), d, d);
  WB_TRY(B.get(p + "/query/bias", 1, &bq, &s));
  WB_REQUIRE(s[0] == d, WB_ERR_SHAPE, "%s/query/bias: expected [%d]", p.c_str(), d);
  WB_TRY(B.get(p + "/value/bias", 1, &bv, &s));
  WB_REQUIRE(s[0] == d, WB_ERR_SHAPE, "%s/value/bias: expected [%d]", p.c_str(), d);
  o->k = d; o->n = 3 * d;
  o->w = B.reserve((size_t)d * 3 * d);
  float* W = &B.host[o->w];
  for (int r = 0; r < d; r++) {
    memcpy(W + (size_t)r * 3 * d, q + (size_t)r * d, sizeof(float) * d);
    memcpy(W + (size_t)r * 3 * d + d, k + (size_t)r * d, sizeof(float) * d);
    memcpy(W + (size_t)r * 3 * d + 2 * d, v + (size_t)r * d, sizeof(float) * d);
  }
  o->b = B.reserve((size_t)3 * d);
  memcpy(&B.host[o->b], bq, sizeof(float) * d);
  memcpy(&B.host[o->b + 2 * d], bv, sizeof(float) * d);
  return WB_OK;
}

static int usize_of(Builder& B, const std::string& name, int* v) {
  float f;
  WB_TRY(B.scalar(name, &f));
  *v = (int)f;   // load.rs:51-53 to_usize
  return WB_OK;
}

int build_model(TensorMap& tm, int device, int compute_dtype, wb_model** out) {
  WB_REQUIRE(compute_dtype != WB_BF16, WB_ERR_ARG,
             "compute_dtype WB_BF16 was retired in round 4: the 16-bit matrix path is the split-precision fp16 kernel of WB_F32 "
             "models (f32-grade results; wb_model_encoder_gemm)");
  WB_REQUIRE(compute_dtype == WB_F32, WB_ERR_ARG, "bad compute_dtype %d", compute_dtype);
  wb::GpuTurn turn(device);   // (the weight repacking / splitting kernels)
  Builder B(tm);
  auto m = std::make_unique<wb_model>();
  wb_dims& D = m->dims;
  // ---- config derivation, load.rs:203-310 ----
  WB_TRY(usize_of(B, "encoder/n_mels", &D.n_mels));
  WB_TRY(usize_of(B, "encoder/n_audio_state", &D.n_audio_state));
  WB_TRY(usize_of(B, "encoder/n_layer", &D.n_audio_layer));
  WB_TRY(usize_of(B, "decoder/n_layer", &D.n_text_layer));
  WB_REQUIRE(D.n_audio_layer >= 1 && D.n_text_layer >= 1, WB_ERR_SHAPE, "n_layer must be >= 1");
  WB_TRY(usize_of(B, "encoder/block_0/attn/n_head", &D.n_audio_head));   // load.rs:229
  WB_TRY(usize_of(B, "decoder/block_0/attn/n_head", &D.n_text_head));    // load.rs:265
  const float* p; std::vector<int64_t> s;
  WB_TRY(B.get("encoder/positional_embedding", 2, &p, &s));
  D.n_audio_ctx = (int)s[0];
  WB_REQUIRE(s[1] == D.n_audio_state, WB_ERR_SHAPE, "encoder/positional_embedding: width %lld != n_audio_state %d",
             (long long)s[1], D.n_audio_state);
  const float* enc_pos_h = p;
  WB_TRY(B.get("decoder/positional_embedding", 2, &p, &s));
  D.n_text_ctx = (int)s[0];
  D.n_text_state = (int)s[1];
  const float* dec_pos_h = p;
  WB_TRY(B.get("decoder/token_embedding/weight", 2, &p, &s));
  D.n_vocab = (int)s[0];
  WB_REQUIRE(s[1] == D.n_text_state, WB_ERR_SHAPE, "token_embedding width != n_text_state");
  const float* emb_h = p;
  // mod.rs:27-32
  WB_REQUIRE(D.n_audio_state == D.n_text_state, WB_ERR_SHAPE,
             "Audio encoder state size %d must be equal to text decoder state size %d.", D.n_audio_state,
             D.n_text_state);
  const int d = D.n_audio_state;
  WB_REQUIRE(D.n_audio_head > 0 && d % D.n_audio_head == 0 && D.n_text_head > 0 && d % D.n_text_head == 0,
             WB_ERR_SHAPE, "State size %d must be a multiple of head size", d);   // mod.rs:393-398
  WB_REQUIRE(d / D.n_audio_head == 64 && d / D.n_text_head == 64, WB_ERR_SHAPE,
             "this engine is built for head size 64 (every Whisper preset); got %d", d / D.n_audio_head);
  WB_REQUIRE(d % 64 == 0 && D.n_mels == 80, WB_ERR_SHAPE, "n_state must be a multiple of 64 and n_mels 80");
  m->qk_scale = (float)std::pow((double)d / (double)D.n_audio_head, -0.25);   // mod.rs:503

  // ---- arena image ----
  Off conv1, conv2;
  {
    const float *w, *b;
    WB_TRY(B.get("encoder/conv1/weight", 3, &w, &s));
    WB_REQUIRE(s[0] == d && s[1] == 80 && s[2] == 3, WB_ERR_SHAPE, "encoder/conv1/weight: expected [%d,80,3]", d);
    WB_TRY(B.get("encoder/conv1/bias", 1, &b, &s));
    WB_REQUIRE(s[0] == d, WB_ERR_SHAPE, "encoder/conv1/bias: expected [%d]", d);
    conv1.k = 240; conv1.n = d;
    conv1.w = B.reserve((size_t)240 * d);
    for (int co = 0; co < d; co++)
      for (int ci = 0; ci < 80; ci++)
        for (int kk = 0; kk < 3; kk++)
          B.host[conv1.w + (size_t)(ci * 3 + kk) * d + co] = w[((size_t)co * 80 + ci) * 3 + kk];
    conv1.b = B.reserve(d); memcpy(&B.host[conv1.b], b, sizeof(float) * d);
    WB_TRY(B.get("encoder/conv2/weight", 3, &w, &s));
    WB_REQUIRE(s[0] == d && s[1] == d && s[2] == 3, WB_ERR_SHAPE, "encoder/conv2/weight: expected [%d,%d,3]", d, d);
    WB_TRY(B.get("encoder/conv2/bias", 1, &b, &s));
    WB_REQUIRE(s[0] == d, WB_ERR_SHAPE, "encoder/conv2/bias: expected [%d]", d);
    conv2.k = 3 * d; conv2.n = d;
    conv2.w = B.reserve((size_t)3 * d * d);
    for (int co = 0; co < d; co++)
      for (int ci = 0; ci < d; ci++)
        for (int kk = 0; kk < 3; kk++)
          B.host[conv2.w + ((size_t)kk * d + ci) * d + co] = w[((size_t)co * d + ci) * 3 + kk];
    conv2.b = B.reserve(d); memcpy(&B.host[conv2.b], b, sizeof(float) * d);
  }
  size_t enc_pos = B.reserve((size_t)D.n_audio_ctx * d);
  memcpy(&B.host[enc_pos], enc_pos_h, sizeof(float) * (size_t)D.n_audio_ctx * d);

  struct EncOff { LnOff ln1, ln2; Off qkv, out, mlp1, mlp2; };
  struct DecOff { LnOff ln1, ln2, ln3; Off qkv, out, cq, cout, mlp1, mlp2; };
  std::vector<EncOff> eo(D.n_audio_layer);
  std::vector<DecOff> dof(D.n_text_layer);
  for (int i = 0; i < D.n_audio_layer; i++) {
    std::string bp = "encoder/block_" + std::to_string(i);
    int nh; WB_TRY(usize_of(B, bp + "/attn/n_head", &nh));
    WB_REQUIRE(nh == D.n_audio_head, WB_ERR_SHAPE, "%s: n_head differs across blocks", bp.c_str());
    WB_TRY(put_ln(B, bp + "/attn_ln", d, &eo[i].ln1));
    WB_TRY(put_qkv(B, bp + "/attn", d, &eo[i].qkv));
    WB_TRY(put_linear(B, bp + "/attn/out", d, d, &eo[i].out, true));
    WB_TRY(put_ln(B, bp + "/mlp_ln", d, &eo[i].ln2));
    WB_TRY(put_linear(B, bp + "/mlp/mlp1", d, 4 * d, &eo[i].mlp1, true));
    WB_TRY(put_linear(B, bp + "/mlp/mlp2", 4 * d, d, &eo[i].mlp2, true));
  }
  LnOff ln_post; WB_TRY(put_ln(B, "encoder/ln_post", d, &ln_post));

  const int V = D.n_vocab, NL = D.n_text_layer;
  size_t emb = B.reserve((size_t)V * d);
  memcpy(&B.host[emb], emb_h, sizeof(float) * (size_t)V * d);
  const int Vp = (V + 63) / 64 * 64;   // E^T rows padded so float4 tile loads never leave the row
  m->vocab_ld = Vp;
  size_t emb_t = B.reserve((size_t)Vp * d);
  for (int v = 0; v < V; v++)
    for (int c = 0; c < d; c++) B.host[emb_t + (size_t)c * Vp + v] = emb_h[(size_t)v * d + c];
  size_t dec_pos = B.reserve((size_t)D.n_text_ctx * d);
  memcpy(&B.host[dec_pos], dec_pos_h, sizeof(float) * (size_t)D.n_text_ctx * d);
  Off ckv; ckv.k = d; ckv.n = NL * 2 * d;
  ckv.w = B.reserve((size_t)d * ckv.n);
  ckv.b = B.reserve((size_t)ckv.n);
  for (int i = 0; i < NL; i++) {
    std::string bp = "decoder/block_" + std::to_string(i);
    int nh; WB_TRY(usize_of(B, bp + "/attn/n_head", &nh));
    WB_REQUIRE(nh == D.n_text_head, WB_ERR_SHAPE, "%s: n_head differs across blocks", bp.c_str());
    WB_TRY(usize_of(B, bp + "/cross_attn/n_head", &nh));
    WB_REQUIRE(nh == D.n_text_head, WB_ERR_SHAPE, "%s: cross n_head differs", bp.c_str());
    WB_TRY(put_ln(B, bp + "/attn_ln", d, &dof[i].ln1));
    WB_TRY(put_qkv(B, bp + "/attn", d, &dof[i].qkv));
    WB_TRY(put_linear(B, bp + "/attn/out", d, d, &dof[i].out, true));
    WB_TRY(put_ln(B, bp + "/cross_attn_ln", d, &dof[i].ln2));
    WB_TRY(put_linear(B, bp + "/cross_attn/query", d, d, &dof[i].cq, true));
    WB_TRY(put_linear(B, bp + "/cross_attn/out", d, d, &dof[i].cout, true));
    WB_TRY(put_ln(B, bp + "/mlp_ln", d, &dof[i].ln3));
    WB_TRY(put_linear(B, bp + "/mlp/mlp1", d, 4 * d, &dof[i].mlp1, true));
    WB_TRY(put_linear(B, bp + "/mlp/mlp2", 4 * d, d, &dof[i].mlp2, true));
    const float *kw, *vw, *vb;
    WB_TRY(B.get(bp + "/cross_attn/key/weight", 2, &kw, &s));
    WB_REQUIRE(s[0] == d && s[1] == d, WB_ERR_SHAPE, "%s/cross_attn/key/weight: expected [%d,%d]", bp.c_str(), d, d);
    WB_TRY(B.get(bp + "/cross_attn/value/weight", 2, &vw, &s));
    WB_REQUIRE(s[0] == d && s[1] == d, WB_ERR_SHAPE, "%s/cross_attn/value/weight: expected [%d,%d]", bp.c_str(), d, d);
    WB_TRY(B.get(bp + "/cross_attn/value/bias", 1, &vb, &s));
    WB_REQUIRE(s[0] == d, WB_ERR_SHAPE, "%s/cross_attn/value/bias: expected [%d]", bp.c_str(), d);
    for (int r = 0; r < d; r++) {
      float* row = &B.host[ckv.w + (size_t)r * ckv.n + (size_t)i * 2 * d];
      memcpy(row, kw + (size_t)r * d, sizeof(float) * d);
      memcpy(row + d, vw + (size_t)r * d, sizeof(float) * d);
    }
    memcpy(&B.host[ckv.b + (size_t)i * 2 * d + d], vb, sizeof(float) * d);
  }
  LnOff ln_dec; WB_TRY(put_ln(B, "decoder/ln", d, &ln_dec));

  // ---- upload ----
  m->device = device;
  m->compute_dtype = compute_dtype;
  WB_HIP(hipSetDevice(device));
  WB_TRY(m->arena.alloc(B.host.size() * sizeof(float)));
  WB_HIP(hipMemcpy(m->arena.p, B.host.data(), B.host.size() * sizeof(float), hipMemcpyHostToDevice));
  WB_HIP(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
  float* base = m->arena.as<float>();
  auto lin = [&](const Off& o) { LinearW l; l.w = base + o.w; l.b = base + o.b; l.k = o.k; l.n = o.n; return l; };
  auto ln = [&](const LnOff& o) { LayerNormW l; l.g = base + o.g; l.b = base + o.b; l.eps = o.eps; return l; };
  m->conv1 = lin(conv1); m->conv2 = lin(conv2);
  m->enc_pos = base + enc_pos;
  m->enc.resize(D.n_audio_layer);
  for (int i = 0; i < D.n_audio_layer; i++) {
    auto& e = m->enc[i];
    e.ln1 = ln(eo[i].ln1); e.ln2 = ln(eo[i].ln2);
    e.qkv = lin(eo[i].qkv); e.out = lin(eo[i].out); e.mlp1 = lin(eo[i].mlp1); e.mlp2 = lin(eo[i].mlp2);
  }
  m->ln_post = ln(ln_post);
  m->tok_emb = base + emb; m->tok_emb_t = base + emb_t; m->dec_pos = base + dec_pos;
  m->dec.resize(NL);
  for (int i = 0; i < NL; i++) {
    auto& e = m->dec[i];
    e.ln1 = ln(dof[i].ln1); e.ln2 = ln(dof[i].ln2); e.ln3 = ln(dof[i].ln3);
    e.qkv = lin(dof[i].qkv); e.out = lin(dof[i].out); e.cq = lin(dof[i].cq); e.cout = lin(dof[i].cout);
    e.mlp1 = lin(dof[i].mlp1); e.mlp2 = lin(dof[i].mlp2);
  }
  m->ckv_all = lin(ckv);
  m->ln_dec = ln(ln_dec);
  // ---- fp16 hi / lo copies of the encoder-side GEMM weights for the three-product kernel
  // (gemm_f16x3.hip: f32-grade results at ~2x the rate of the exact-f32 MFMA); made on the device, once.
  // WHISPER_HIP_ENCODER_SPLIT=0 keeps the exact-f32 MFMA kernel for everything.
  static const bool split_enabled = []() { const char* e = getenv("WHISPER_HIP_ENCODER_SPLIT"); return e ? e[0] == '1' : WB_ENCODER_SPLIT_DEFAULT; }();
  if (split_enabled) {
    std::vector<LinearW*> cand, ws;
    for (int i = 0; i < D.n_audio_layer; i++) {
      cand.push_back(&m->enc[i].qkv); cand.push_back(&m->enc[i].out); cand.push_back(&m->enc[i].mlp1); cand.push_back(&m->enc[i].mlp2);
    }
    cand.push_back(&m->ckv_all);
    cand.push_back(&m->conv2);            // K = 3 d: the longest f32 chain of the encoder otherwise (masks are multiples of d)
    // a weight outside fp16's range keeps the exact-f32 kernel (its hi piece would be inf); weights live in the host image
    const float* arena_dev = m->arena.as<float>();
    for (LinearW* l : cand) {
      const float* hw = B.host.data() + (l->w - arena_dev);
      bool ok = true;
      for (size_t i = 0, n = (size_t)l->k * l->n; i < n && ok; i++) ok = std::fabs(hw[i]) < 65000.f;
      if (ok) ws.push_back(l);
    }
    size_t total = 0;
    for (LinearW* l : ws) total += ((size_t)l->k * l->n + 127) & ~size_t(127);
    if (total) {                                // (no weight qualifies: no flag word, no arena -- the model is exact-f32)
      WB_HIP(hipHostMalloc((void**)&m->split_flag_host, 64, hipHostMallocMapped));      // freed by ~wb_model
      *m->split_flag_host = 0;
      WB_HIP(hipHostGetDevicePointer((void**)&m->split_flag_dev, m->split_flag_host, 0));
      WB_TRY(m->arena_split.alloc(total * 2 * 2));
    }
    uint16_t* p = m->arena_split.as<uint16_t>();
    for (LinearW* l : ws) {
      const size_t n = ((size_t)l->k * l->n + 127) & ~size_t(127);
      l->sh = p; l->sl = p + n; p += 2 * n;
      launch_split_weight_f16(nullptr, l->w, l->k, l->n, l->sh, l->sl);
      WB_HIP(hipGetLastError());
    }
    WB_HIP(hipDeviceSynchronize());
  }
  // ---- fp16 hi / lo TILES of the decoder-side Linear weights for the batch-mode skinny GEMM (decode_batch.hip,
  // dec_skinny_f16x3_kernel: the decoder's weight stream on the 16-bit matrix path, f32-grade results).  The f32 copies stay:
  // the <= 8-row kernels (fused sublayers, persistent decode) and the stateless decoder use them.
  // WHISPER_HIP_DECODER_SPLIT=0 keeps the exact-f32 skinny kernel.
  static const bool dec_split_enabled = []() { const char* e = getenv("WHISPER_HIP_DECODER_SPLIT"); return !(e && e[0] == '0'); }();
  if (dec_split_enabled && compute_dtype == WB_F32) {
    std::vector<LinearW*> ws;
    const float* arena_dev = m->arena.as<float>();
    for (int i = 0; i < D.n_text_layer; i++)
      for (LinearW* l : {&m->dec[i].qkv, &m->dec[i].out, &m->dec[i].cq, &m->dec[i].cout, &m->dec[i].mlp1, &m->dec[i].mlp2}) {
        if (l->k % 32 != 0 || l->n % 64 != 0) continue;
        const float* hw = B.host.data() + (l->w - arena_dev);
        bool ok = true;
        for (size_t j = 0, n = (size_t)l->k * l->n; j < n && ok; j++) ok = std::fabs(hw[j]) < 65000.f;
        if (ok) ws.push_back(l);
      }
    size_t total = 0;
    for (LinearW* l : ws) total += (size_t)l->k * l->n;
    // all or nothing: one arithmetic per decode step.  The tiles cost another 4 B per decoder-layer weight element (2.9 GB at
    // large-v2, next to 6.5 GB of f32 weights): when the device cannot spare them the model simply stays on the exact-f32
    // skinny kernel -- an optional speed path must not fail the load (wb_model_decoder_gemm then reports 0)
    if (total && ws.size() == (size_t)D.n_text_layer * 6 && m->arena_dec_split.alloc(total * 2 * 2) != WB_OK) {
      (void)hipGetLastError();                                       // (clear the sticky out-of-memory status)
      ws.clear();
    }
    if (total && ws.size() == (size_t)D.n_text_layer * 6) {
      uint16_t* p = m->arena_dec_split.as<uint16_t>();
      for (LinearW* l : ws) {
        const size_t n = (size_t)l->k * l->n;
        l->th = p; l->tl = p + n; p += 2 * n;
        launch_split_weight_f16_tiles(nullptr, l->w, l->k, l->n, l->th, l->tl);
        WB_HIP(hipGetLastError());
      }
      WB_HIP(hipDeviceSynchronize());
    }
  }
  *out = m.release();
  session_pool_register(*out);
  return WB_OK;
}

}  // namespace wb
