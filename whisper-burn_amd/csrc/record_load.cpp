// Reader for the reference's converted model files: `<name>.mpk.gz` (+ optional `<name>.cfg`).
//
// The reference converts a dump directory to a Burn record with
//   NamedMpkGzFileRecorder::<FullPrecisionSettings>   (/root/reference/src/bin/convert/main.rs:17-19, :51)
// and loads it back through the same recorder (src/bin/transcribe/main.rs:63-70, :126).  That is
//   gzip( MessagePack, structs as maps with field names (rmp-serde "named") )
// of  { metadata: {...}, item: <the Whisper module record> }  where the module record mirrors the module
// tree of src/model/mod.rs (:42-45, :120-128, :215-225, :291-296, :335-342, :369-373, :419-425, :473-479), a
// `Vec<Block>` is an array, and every `Param<Tensor>` is a map  { id, param: { .. value: [f32...], shape: [usize...] } }.
// Burn 0.9.0 (git fb2a71bb) is not vendored in the reference tree, so the exact nesting under `param` cannot be
// checked here: the walker is structural -- ANY map holding a numeric array "value" and an integer array
// "shape" is a tensor, named by the path of keys that leads to it (wrapper keys item / param / data / tensor are
// transparent).  Format status: parity unpinned (no reference-written file exists offline); tests write records
// in this layout with Python msgpack.
//
// Name mapping to the dump-directory names the rest of the loader uses (load.rs:19-200):
//   blocks[i] -> block_i, lin1 / lin2 -> mlp1 / mlp2, LayerNorm gamma / beta -> weight / bias (epsilon field or
//   1e-5), decoder.token_embedding -> decoder/token_embedding/weight, decoder.mask ignored (implicit causal mask);
//   head counts from the .cfg JSON when given, else n_state / 64.
#include <zlib.h>

#include <cstring>
#include <functional>

#include "wb_internal.h"

using namespace wb;

namespace {

struct Node {
  enum Kind { NIL, BOOL, INT, FLT, STR, ARR, MAP, NUMARR } kind = NIL;
  int64_t i = 0;
  double f = 0.0;
  std::string s;
  std::vector<Node> arr;                                   // ARR
  std::vector<std::pair<std::string, Node>> map;           // MAP (string keys; other keys are stringified)
  std::vector<float> nums;                                 // NUMARR: an array of numbers, kept flat
  bool nums_all_int = true;
};

struct Parser {
  const unsigned char* p; const unsigned char* end; std::string err;
  bool need(size_t n) { if ((size_t)(end - p) < n) { err = "truncated MessagePack stream"; return false; } return true; }
  uint64_t be(int n) { uint64_t v = 0; for (int k = 0; k < n; k++) v = (v << 8) | p[k]; p += n; return v; }

  bool number(unsigned char t, double* out, bool* is_int) {   // after the type byte
    *is_int = true;
    if (t <= 0x7f) { *out = t; return true; }
    if (t >= 0xe0) { *out = (int8_t)t; return true; }
    switch (t) {
      case 0xcc: if (!need(1)) return false; *out = (double)be(1); return true;
      case 0xcd: if (!need(2)) return false; *out = (double)be(2); return true;
      case 0xce: if (!need(4)) return false; *out = (double)be(4); return true;
      case 0xcf: if (!need(8)) return false; *out = (double)be(8); return true;
      case 0xd0: if (!need(1)) return false; *out = (double)(int8_t)be(1); return true;
      case 0xd1: if (!need(2)) return false; *out = (double)(int16_t)be(2); return true;
      case 0xd2: if (!need(4)) return false; *out = (double)(int32_t)be(4); return true;
      case 0xd3: if (!need(8)) return false; *out = (double)(int64_t)be(8); return true;
      case 0xca: { if (!need(4)) return false; uint32_t u = (uint32_t)be(4); float f; memcpy(&f, &u, 4); *out = f; *is_int = false; return true; }
      case 0xcb: { if (!need(8)) return false; uint64_t u = be(8); double d; memcpy(&d, &u, 8); *out = d; *is_int = false; return true; }
      default: return false;
    }
  }
  static bool is_number_tag(unsigned char t) { return t <= 0x7f || t >= 0xe0 || (t >= 0xca && t <= 0xd3); }

  bool str(size_t n, std::string* out) { if (!need(n)) return false; out->assign((const char*)p, n); p += n; return true; }

  bool array(size_t n, Node* out) {
    if (n > 0 && need(1) && is_number_tag(*p)) {            // a flat run of numbers: no node per element
      out->kind = Node::NUMARR;
      out->nums.resize(n);
      for (size_t k = 0; k < n; k++) {
        if (!need(1)) return false;
        const unsigned char t = *p++;
        double v; bool ii;
        if (!is_number_tag(t) || !number(t, &v, &ii)) { if (err.empty()) err = "mixed array of numbers and objects"; return false; }
        out->nums[k] = (float)v;
        out->nums_all_int = out->nums_all_int && ii;
      }
      return true;
    }
    out->kind = Node::ARR;
    out->arr.resize(n);
    for (size_t k = 0; k < n; k++) if (!value(&out->arr[k])) return false;
    return true;
  }
  bool map(size_t n, Node* out) {
    out->kind = Node::MAP;
    out->map.resize(n);
    for (size_t k = 0; k < n; k++) {
      Node key;
      if (!value(&key)) return false;
      out->map[k].first = key.kind == Node::STR ? key.s : std::to_string(key.i);
      if (!value(&out->map[k].second)) return false;
    }
    return true;
  }
  bool value(Node* out) {
    if (!need(1)) return false;
    const unsigned char t = *p++;
    if (is_number_tag(t)) {
      double v; bool ii;
      if (!number(t, &v, &ii)) return false;
      out->kind = ii ? Node::INT : Node::FLT; out->i = (int64_t)v; out->f = v;
      return true;
    }
    if ((t & 0xe0) == 0xa0) { out->kind = Node::STR; return str(t & 0x1f, &out->s); }
    if ((t & 0xf0) == 0x90) return array(t & 0x0f, out);
    if ((t & 0xf0) == 0x80) return map(t & 0x0f, out);
    switch (t) {
      case 0xc0: out->kind = Node::NIL; return true;
      case 0xc2: case 0xc3: out->kind = Node::BOOL; out->i = t == 0xc3; return true;
      case 0xd9: if (!need(1)) return false; out->kind = Node::STR; return str((size_t)be(1), &out->s);
      case 0xda: if (!need(2)) return false; out->kind = Node::STR; return str((size_t)be(2), &out->s);
      case 0xdb: if (!need(4)) return false; out->kind = Node::STR; return str((size_t)be(4), &out->s);
      case 0xc4: if (!need(1)) return false; out->kind = Node::STR; return str((size_t)be(1), &out->s);   // bin: kept as bytes
      case 0xc5: if (!need(2)) return false; out->kind = Node::STR; return str((size_t)be(2), &out->s);
      case 0xc6: if (!need(4)) return false; out->kind = Node::STR; return str((size_t)be(4), &out->s);
      case 0xdc: if (!need(2)) return false; return array((size_t)be(2), out);
      case 0xdd: if (!need(4)) return false; return array((size_t)be(4), out);
      case 0xde: if (!need(2)) return false; return map((size_t)be(2), out);
      case 0xdf: if (!need(4)) return false; return map((size_t)be(4), out);
      default: err = "unsupported MessagePack type (ext)"; return false;
    }
  }
};

const Node* find(const Node& m, const char* key) {
  if (m.kind != Node::MAP) return nullptr;
  for (auto& kv : m.map) if (kv.first == key) return &kv.second;
  return nullptr;
}

bool transparent(const std::string& k) { return k == "item" || k == "param" || k == "data" || k == "tensor"; }

std::string rename(const std::string& k) {
  if (k == "lin1") return "mlp1";
  if (k == "lin2") return "mlp2";
  if (k == "gamma") return "weight";
  if (k == "beta") return "bias";
  return k;
}

// depth-first walk: tensors and LayerNorm epsilons, named by their path
void walk(Node& n, const std::string& path, TensorMap* out) {
  if (n.kind == Node::MAP) {
    const Node* v = find(n, "value");
    const Node* sh = find(n, "shape");
    if (v && sh && v->kind == Node::NUMARR && (sh->kind == Node::NUMARR || (sh->kind == Node::ARR && sh->arr.empty()))) {
      HostTensor t;
      if (sh->kind == Node::NUMARR) for (float d : sh->nums) t.shape.push_back((int64_t)d);
      for (auto& kv : n.map) if (kv.first == "value") { t.owned = std::move(kv.second.nums); break; }
      t.data = nullptr;
      (*out)[path] = std::move(t);
      return;
    }
    for (auto& kv : n.map) {
      if (kv.first == "epsilon" && (kv.second.kind == Node::FLT || kv.second.kind == Node::INT)) {
        HostTensor t; t.shape = {1}; t.owned = {(float)kv.second.f};
        (*out)[path + "/eps"] = std::move(t);
        continue;
      }
      const std::string sub = transparent(kv.first) ? path : (path.empty() ? rename(kv.first) : path + "/" + rename(kv.first));
      walk(kv.second, sub, out);
    }
  } else if (n.kind == Node::ARR) {
    // `blocks: Vec<Block>` -> block_<i>; any other array of records keeps <name>_<i>
    std::string base = path;
    const size_t cut = base.rfind('/');
    std::string leaf = cut == std::string::npos ? base : base.substr(cut + 1);
    if (leaf == "blocks") leaf = "block";
    const std::string parent = cut == std::string::npos ? "" : base.substr(0, cut + 1);
    for (size_t i = 0; i < n.arr.size(); i++) walk(n.arr[i], parent + leaf + "_" + std::to_string(i), out);
  }
}

int inflate_file(const char* path, std::vector<unsigned char>* out) {
  gzFile g = gzopen(path, "rb");
  WB_REQUIRE(g, WB_ERR_IO, "cannot open %s", path);
  std::vector<unsigned char> buf((size_t)1 << 22);
  for (;;) {
    const int n = gzread(g, buf.data(), (unsigned)buf.size());
    if (n < 0) { gzclose(g); WB_REQUIRE(false, WB_ERR_IO, "%s: gzip stream is damaged", path); }
    if (n == 0) break;
    out->insert(out->end(), buf.begin(), buf.begin() + n);
  }
  gzclose(g);
  return WB_OK;
}

// "key": <integer> inside the WhisperConfig JSON (src/model/mod.rs:16-20, :73-80, :164-171); -1 if absent
int json_int(const std::string& js, const char* key) {
  const std::string k = std::string("\"") + key + "\"";
  size_t at = js.find(k);
  if (at == std::string::npos) return -1;
  at = js.find(':', at + k.size());
  if (at == std::string::npos) return -1;
  return atoi(js.c_str() + at + 1);
}

void put_scalar(TensorMap* tm, const std::string& name, float v) {
  HostTensor t; t.shape = {1}; t.owned = {v};
  (*tm)[name] = std::move(t);
}

}  // namespace

namespace wb {

int read_burn_record(const char* path, const char* cfg_path, TensorMap* tm) {
  std::vector<unsigned char> raw;
  WB_TRY(inflate_file(path, &raw));
  WB_REQUIRE(!raw.empty(), WB_ERR_IO, "%s: empty record", path);
  Parser ps{raw.data(), raw.data() + raw.size(), {}};
  Node root;
  WB_REQUIRE(ps.value(&root), WB_ERR_IO, "%s: %s", path, ps.err.empty() ? "not a MessagePack record" : ps.err.c_str());
  raw.clear(); raw.shrink_to_fit();
  WB_REQUIRE(root.kind == Node::MAP, WB_ERR_IO, "%s: the record is not a named (map) MessagePack struct", path);
  if (const Node* md = find(root, "metadata")) {
    const Node* fl = find(*md, "float");
    WB_REQUIRE(!fl || fl->kind != Node::STR || fl->s == "f32" || fl->s == "f64", WB_ERR_IO,
               "%s: record precision '%s' not supported (FullPrecisionSettings f32 expected, convert/main.rs:51)", path,
               fl ? fl->s.c_str() : "?");
  }
  walk(root, "", tm);
  WB_REQUIRE(tm->count("encoder/conv1/weight") && tm->count("decoder/positional_embedding"), WB_ERR_IO,
             "%s: no Whisper module record found (encoder.conv1.weight / decoder.positional_embedding missing).  Layout "
             "assumed (Burn 0.9.0 @ fb2a71bb is not vendored with the reference, so it could not be checked against a "
             "converter-written file): gzip(named MessagePack) of {metadata, item}, item = the module tree of "
             "src/model/mod.rs in field order (encoder{conv1, gelu1, conv2, gelu2, blocks[], ln_post, positional_embedding, "
             "n_mels, n_audio_ctx}, decoder{token_embedding, positional_embedding, blocks[], ln, mask, n_vocab, n_text_ctx}), "
             "a Param as {id, param: {value[], shape[]}} or {id, param: {data: {value[], shape[]}}}, Linear {weight, bias}, "
             "LayerNorm {gamma, beta, epsilon}", path);
  tm->erase("decoder/mask");                               // mod.rs:125: a stored 448 x 448 mask; causality is implicit here
  if (tm->count("decoder/token_embedding")) {              // a bare Param (mod.rs:121) -> the dump's <name>/weight
    (*tm)["decoder/token_embedding/weight"] = std::move((*tm)["decoder/token_embedding"]);
    tm->erase("decoder/token_embedding");
  }
  // scalars the dump directory carries explicitly (load.rs:203-310)
  int n_enc = 0, n_dec = 0;
  while (tm->count("encoder/block_" + std::to_string(n_enc) + "/attn/query/weight")) n_enc++;
  while (tm->count("decoder/block_" + std::to_string(n_dec) + "/attn/query/weight")) n_dec++;
  WB_REQUIRE(n_enc > 0 && n_dec > 0, WB_ERR_IO, "%s: no encoder / decoder blocks in the record", path);
  const HostTensor& c1 = (*tm)["encoder/conv1/weight"];
  WB_REQUIRE(c1.shape.size() == 3, WB_ERR_SHAPE, "encoder/conv1/weight: rank %zu", c1.shape.size());
  const int d = (int)c1.shape[0];
  int h_enc = -1, h_dec = -1;
  if (cfg_path && cfg_path[0]) {
    FILE* f = fopen(cfg_path, "rb");
    WB_REQUIRE(f, WB_ERR_IO, "cannot open %s", cfg_path);
    std::string js; char b[4096]; size_t k;
    while ((k = fread(b, 1, sizeof(b), f)) > 0) js.append(b, k);
    fclose(f);
    h_enc = json_int(js, "n_audio_head"); h_dec = json_int(js, "n_text_head");
  }
  if (h_enc <= 0) h_enc = d / 64;                           // every Whisper preset has head size 64
  if (h_dec <= 0) h_dec = d / 64;
  put_scalar(tm, "encoder/n_mels", (float)c1.shape[1]);
  put_scalar(tm, "encoder/n_audio_state", (float)d);
  put_scalar(tm, "encoder/n_layer", (float)n_enc);
  put_scalar(tm, "decoder/n_layer", (float)n_dec);
  for (int i = 0; i < n_enc; i++) put_scalar(tm, "encoder/block_" + std::to_string(i) + "/attn/n_head", (float)h_enc);
  for (int i = 0; i < n_dec; i++) {
    put_scalar(tm, "decoder/block_" + std::to_string(i) + "/attn/n_head", (float)h_dec);
    put_scalar(tm, "decoder/block_" + std::to_string(i) + "/cross_attn/n_head", (float)h_dec);
  }
  // LayerNorm epsilon: the record's `epsilon` field when present, else Burn's LayerNormConfig default
  std::vector<std::string> lns;
  for (auto& kv : *tm) {
    const std::string& nm = kv.first;
    const bool is_ln = nm.size() > 7 && nm.compare(nm.size() - 7, 7, "/weight") == 0 &&
                       (nm.find("_ln/") != std::string::npos || nm.find("/ln/") != std::string::npos ||
                        nm.find("/ln_post/") != std::string::npos);
    if (is_ln) lns.push_back(nm.substr(0, nm.size() - 7));
  }
  for (auto& b : lns) if (!tm->count(b + "/eps")) put_scalar(tm, b + "/eps", 1e-5f);
  for (auto& kv : *tm) if (!kv.second.data) kv.second.data = kv.second.owned.data();
  return WB_OK;
}

}  // namespace wb

extern "C" {

int wb_burn_record_read(const char* mpk_gz_path, const char* cfg_path, wb_tensor_fn fn, void* user) {
  WB_REQUIRE(mpk_gz_path && fn, WB_ERR_ARG, "wb_burn_record_read: null argument");
  TensorMap tm;
  WB_TRY(read_burn_record(mpk_gz_path, cfg_path, &tm));
  for (auto& kv : tm) {
    const int rc = fn(user, kv.first.c_str(), kv.second.data, kv.second.shape.data(), (int32_t)kv.second.shape.size());
    if (rc != 0) return rc;
  }
  return WB_OK;
}

int wb_model_load_burn_record(const char* mpk_gz_path, const char* cfg_path, int device, int compute_dtype,
                              wb_model** out) {
  WB_REQUIRE(mpk_gz_path && out, WB_ERR_ARG, "wb_model_load_burn_record: null argument");
  TensorMap tm;
  WB_TRY(read_burn_record(mpk_gz_path, cfg_path, &tm));
  return build_model(tm, device, compute_dtype, out);
}

}  // extern "C"
