// Developer probe (not part of the library): per-kernel latency of the decode-step kernels at
// tiny.en geometry, launched back to back, warm vs rotating (cold) weights, eager vs hipGraph.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/decode_probe.cpp csrc/build/decode.hip.o <decode_fused.hip built with -DWB_STAMPS> -o tools/decode_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <functional>
#include <vector>
#include "../csrc/decode.h"
using namespace wb;
namespace wb { void set_error(const char*, ...) {} void prof_tag(int, double) {} bool prof_take_events(hipEvent_t*, hipEvent_t*) { return false; } }
__global__ void k_empty() {}
static hipStream_t st;
static double bench(int reps, const std::function<void(int)>& f, bool graph) {
  if (graph) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 100; i++) f(i);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int r = 0; r < reps / 100; r++) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    auto t1 = std::chrono::high_resolution_clock::now();
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / (reps / 100 * 100);
  }
  for (int i = 0; i < 20; i++) f(i);
  hipStreamSynchronize(st);
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < reps; i++) f(i);
  hipStreamSynchronize(st);
  auto t1 = std::chrono::high_resolution_clock::now();
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
}
int main(int argc, char** argv) {
  const int d = argc > 1 ? atoi(argv[1]) : 384, H = d / 64, n = argc > 2 ? atoi(argv[2]) : 3, S = n, V = 51864, Vp = (V + 63) / 64 * 64;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  const int NCOPY = 48;
  auto dmalloc = [](size_t bytes) { void* p; hipMalloc(&p, bytes); hipMemset(p, 0, bytes); return p; };
  float* Wdd = (float*)dmalloc((size_t)NCOPY * d * d * 4);
  float* Wqkv = (float*)dmalloc((size_t)NCOPY * d * 3 * d * 4);
  float* W1 = (float*)dmalloc((size_t)NCOPY * d * 4 * d * 4);
  float* Et = (float*)dmalloc((size_t)d * Vp * 4);
  float* x = (float*)dmalloc((size_t)2 * S * d * 4);
  float* P = (float*)dmalloc((size_t)16 * S * 4 * d * 4);
  float* P2 = (float*)dmalloc((size_t)16 * S * 4 * d * 4);
  float* bias = (float*)dmalloc(4 * d * 4);
  float* g = (float*)dmalloc(d * 4); float* b = (float*)dmalloc(d * 4);
  float* logits = (float*)dmalloc((size_t)S * V * 4);
  float* tstats = (float*)dmalloc((size_t)S * 1024 * TS_STRIDE * 4);
  float* mask = (float*)dmalloc(V * 4);
  StepLayout L = make_step_layout(S, S);
  std::vector<int> hs(L.total, 0);
  hs[ST_N] = n; hs[ST_STEP] = 0;
  for (int i = 0; i < n; i++) { hs[L.tok + i] = 5; hs[L.parent + i] = -1; hs[L.len + i] = 50; hs[L.win + i] = i; hs[L.win_nb + i] = 1; hs[L.win_slots + i * MAX_BEAMS] = i; }
  int* stdev = (int*)dmalloc(L.total * 4);
  hipMemcpy(stdev, hs.data(), L.total * 4, hipMemcpyHostToDevice);
  const int Lmax = 128;
  int* tabs = (int*)dmalloc((size_t)2 * S * Lmax * 4);
  std::vector<int> ht(2 * S * Lmax);
  for (size_t i = 0; i < ht.size(); i++) ht[i] = (int)(i % (S * Lmax));
  hipMemcpy(tabs, ht.data(), ht.size() * 4, hipMemcpyHostToDevice);
  float* Kc = (float*)dmalloc((size_t)S * Lmax * d * 4); float* Vc = (float*)dmalloc((size_t)S * Lmax * d * 4);
  float* att = (float*)dmalloc((size_t)S * d * 4);
  const int C = 750, NL = 4, ldkv = NL * 2 * d, nch = (C + 127) / 128;
  float* ckv = (float*)dmalloc((size_t)S * C * ldkv * 4);
  std::vector<int> meta(2 * S);
  for (int w = 0; w < S; w++) { meta[w] = w * C; meta[S + w] = C; }
  int* wmeta = (int*)dmalloc(meta.size() * 4);
  hipMemcpy(wmeta, meta.data(), meta.size() * 4, hipMemcpyHostToDevice);
  float* ca = (float*)dmalloc((size_t)S * H * nch * CA_STRIDE * 4);
  int ks_o, ksl_o, ks_q, ksl_q, ks_1, ksl_1;
  gemv_plan(d, d, &ks_o, &ksl_o); gemv_plan(d, 3 * d, &ks_q, &ksl_q); gemv_plan(d, 4 * d, &ks_1, &ksl_1);
  printf("d=%d n=%d plans: o ks=%d ksl=%d | qkv ks=%d ksl=%d | mlp1 ks=%d ksl=%d\n", d, n, ks_o, ksl_o, ks_q, ksl_q, ks_1, ksl_1);
  auto base = [&](const float* W, int K, int N, int ks, int ksl) {
    GemvArgs a; a.W = W; a.ldw = N; a.K = K; a.N = N; a.KS = ks; a.KSL = ksl; a.P = P2; a.st = stdev; a.S = S; return a;
  };
  struct Case { const char* name; std::function<void(int)> f; };
  std::vector<Case> cases;
  cases.push_back({"empty kernel (3 blocks)", [&](int) { hipLaunchKernelGGL(k_empty, dim3(3), dim3(256), 0, st); }});
  for (int cold = 0; cold < 2; cold++) {
    cases.push_back({cold ? "gemv plain Wo [d,d] cold" : "gemv plain Wo [d,d] warm", [&, cold](int i) {
      GemvArgs a = base(Wdd + (size_t)(cold ? i % NCOPY : 0) * d * d, d, d, ks_o, ksl_o); a.pro = PRO_PLAIN; a.src = att; a.ld_src = d;
      launch_dec_gemv(st, a, n, false); }});
    cases.push_back({cold ? "gemv LN qkv [d,3d] cold" : "gemv LN qkv [d,3d] warm", [&, cold](int i) {
      GemvArgs a = base(Wqkv + (size_t)(cold ? i % NCOPY : 0) * d * 3 * d, d, 3 * d, ks_q, ksl_q); a.pro = PRO_LN; a.src = x; a.ld_src = d;
      a.pend = P; a.KSp = ks_o; a.pbias = bias; a.x_out = x + (size_t)S * d; a.ln_g = g; a.ln_b = b; a.ln_eps = 1e-5f;
      launch_dec_gemv(st, a, n, false); }});
    cases.push_back({cold ? "gemv LN mlp1 [d,4d] cold" : "gemv LN mlp1 [d,4d] warm", [&, cold](int i) {
      GemvArgs a = base(W1 + (size_t)(cold ? i % NCOPY : 0) * d * 4 * d, d, 4 * d, ks_1, ksl_1); a.pro = PRO_LN; a.src = x; a.ld_src = d;
      a.pend = P; a.KSp = ks_o; a.pbias = bias; a.x_out = x + (size_t)S * d; a.ln_g = g; a.ln_b = b; a.ln_eps = 1e-5f;
      launch_dec_gemv(st, a, n, false); }});
  }
  float* W2m = (float*)dmalloc((size_t)NCOPY * 4 * d * d * 4);
  float* Pa = (float*)dmalloc((size_t)64 * S * d * 4);
  for (int cold = 0; cold < 2; cold++) {
    cases.push_back({cold ? "FUSED mlp block cold" : "FUSED mlp block warm", [&, cold](int i) {
      MlpFusedArgs ma; ma.st = stdev; ma.S = S; ma.d = d; ma.x_in = x; ma.pend = P; ma.KSp = ks_o; ma.pbias = bias; ma.x_out = x + (size_t)S * d;
      ma.ln_g = g; ma.ln_b = b; ma.ln_eps = 1e-5f; ma.W1 = W1 + (size_t)(cold ? i % NCOPY : 0) * d * 4 * d; ma.ld1 = 4 * d; ma.b1 = bias;
      ma.W2 = W2m + (size_t)(cold ? i % NCOPY : 0) * 4 * d * d; ma.P = Pa;
      launch_dec_mlp_fused(st, ma, n); }});
    cases.push_back({cold ? "FUSED attn block cold" : "FUSED attn block warm", [&, cold](int i) {
      AttnFusedArgs fa; fa.st = stdev; fa.lay = L; fa.S = S; fa.d = d; fa.n_head = H; fa.x_in = x; fa.pend = Pa; fa.KSp = 4 * d / 64; fa.pbias = bias;
      fa.x_out = x + (size_t)S * d; fa.ln_g = g; fa.ln_b = b; fa.ln_eps = 1e-5f; fa.Wqkv = Wqkv + (size_t)(cold ? i % NCOPY : 0) * d * 3 * d; fa.ldqkv = 3 * d;
      fa.bqkv = bias; fa.scale = 0.35f; fa.Kc = Kc; fa.Vc = Vc; fa.tabs = tabs; fa.Lmax = Lmax; fa.Wo = Wdd + (size_t)(cold ? i % NCOPY : 0) * d * d; fa.P = P2;
      launch_dec_attn_fused(st, fa, n); }});
  }
  cases.push_back({"self-attn len 50", [&](int) { launch_dec_self_attn(st, stdev, L, n, H, P, ks_q, bias, d, Kc, Vc, tabs, Lmax, 0.35f, att); }});
  cases.push_back({"cross-attn C 750", [&](int) { launch_dec_cross_attn(st, stdev, L, S, H, nch, P, ks_o, bias, d, ckv, ldkv, 0, wmeta, wmeta + S, 0.35f, ca, 1); }});
  cases.push_back({"logits + stats", [&](int) {
    GemvArgs a = base(Et, d, V, 1, d); a.ldw = Vp; a.P = logits; a.pro = PRO_LN; a.src = x; a.ld_src = d; a.pend = P; a.KSp = ks_o; a.pbias = bias;
    a.x_out = x + (size_t)S * d; a.ln_g = g; a.ln_b = b; a.ln_eps = 1e-5f; a.mask = mask; a.topk = 1; a.tstats = tstats;
    launch_dec_gemv(st, a, n, true); }});
  cases.push_back({"topk merge", [&](int) { launch_dec_topk_merge(st, stdev, n, tstats, (V + GV_CT_LOGITS - 1) / GV_CT_LOGITS, 1, (int32_t*)att, att + 64, att + 128, L, nullptr, nullptr, Lmax, -1, NextPrep()); }});
  {
    unsigned long long* stamps = (unsigned long long*)dmalloc(16 * 8);
    unsigned long long hs2[16];
    auto timeline = [&](const char* name, int n_st, const std::function<void()>& f) {
      for (int rep = 0; rep < 3; rep++) { f(); hipStreamSynchronize(st); }
      hipMemcpy(hs2, stamps, 16 * 8, hipMemcpyDeviceToHost);
      printf("%s phase timeline (us since block start):", name);
      for (int i = 1; i < n_st; i++) printf(" %.2f", (double)(hs2[i] - hs2[0]) * 0.01);
      printf("\n");
    };
    timeline("FUSED mlp ", 5, [&]() {
      MlpFusedArgs ma; ma.st = stdev; ma.S = S; ma.d = d; ma.x_in = x; ma.pend = P; ma.KSp = ks_o; ma.pbias = bias; ma.x_out = x + (size_t)S * d;
      ma.ln_g = g; ma.ln_b = b; ma.ln_eps = 1e-5f; ma.W1 = W1; ma.ld1 = 4 * d; ma.b1 = bias; ma.W2 = W2m; ma.P = Pa; ma.stamps = stamps;
      launch_dec_mlp_fused(st, ma, n); });
    timeline("FUSED attn", 8, [&]() {
      AttnFusedArgs fa; fa.st = stdev; fa.lay = L; fa.S = S; fa.d = d; fa.n_head = H; fa.x_in = x; fa.pend = Pa; fa.KSp = 4 * d / 64; fa.pbias = bias;
      fa.x_out = x + (size_t)S * d; fa.ln_g = g; fa.ln_b = b; fa.ln_eps = 1e-5f; fa.Wqkv = Wqkv; fa.ldqkv = 3 * d;
      fa.bqkv = bias; fa.scale = 0.35f; fa.Kc = Kc; fa.Vc = Vc; fa.tabs = tabs; fa.Lmax = Lmax; fa.Wo = Wdd; fa.P = P2; fa.stamps = stamps;
      launch_dec_attn_fused(st, fa, n); });
  }
  for (auto& c : cases) {
    double e = bench(1000, c.f, false), gph = bench(1000, c.f, true);
    printf("%-28s eager %6.2f us   graph %6.2f us\n", c.name, e, gph);
  }
  return 0;
}
