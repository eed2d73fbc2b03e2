// hipemu runtime: fibers, the block scheduler, the wave collectives and the host API subset the engine uses.
// See hip/hip_runtime.h for what is modelled and what is not.  TEST INFRASTRUCTURE -- never part of the product.
#include <hip/hip_runtime.h>
#include <execinfo.h>
#include <pthread.h>
#include <semaphore.h>
#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <mutex>
#include <unordered_map>
#include <vector>

struct ihipStream_t { bool capturing = false; ihipGraph* graph = nullptr; };
struct ihipEvent_t { double t_ms = 0; };
struct ihipGraph { std::vector<std::function<void()>> nodes; };
struct ihipGraphExec { std::vector<std::function<void()>> nodes; };

namespace hipemu {

enum { RUNNABLE = 0, AT_BARRIER = 1, AT_WAVE = 2, DONE = 3, AT_SPIN = 4 };
constexpr size_t STACK_BYTES = 256 << 10;

// A fiber switch is the callee-saved register file and the stack pointer (glibc's swapcontext makes a signal-mask
// system call per switch -- three orders of magnitude slower).
struct Ctx { void* sp = nullptr; };
extern "C" void hipemu_switch(Ctx* from, Ctx* to);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  subq $8, %rsp
  stmxcsr (%rsp)
  fnstcw 4(%rsp)
  movq %rsp, (%rdi)
  movq (%rsi), %rsp
  ldmxcsr (%rsp)
  fldcw 4(%rsp)
  addq $8, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hipemu_switch, .-hipemu_switch
)");

struct Fiber {
  Ctx ctx;
  char* stack = nullptr;
  ThreadCtx tc;
  int state = DONE;
  // mailbox of the pending wave collective
  int op = 0; const void* in0 = nullptr; const void* in1 = nullptr; const void* in2 = nullptr; void* out = nullptr;
  int imm[4] = {0, 0, 0, 0};
};

// Scheduler state is per OS thread: an ordinary launch runs on the caller's thread, a co-resident launch runs every
// block on a thread of its own (so that the thread-local __shared__ statics are per block).
thread_local ThreadCtx* g_tc = nullptr;
static thread_local Fiber* g_cur = nullptr;
static thread_local Ctx g_sched;
static thread_local std::vector<Fiber*> g_fibers;
static thread_local const std::function<void()>* g_body = nullptr;
static thread_local std::vector<char> g_dyn;
static std::recursive_mutex g_mu;
static ihipStream_t g_null_stream;

// ---- co-resident launches: one thread per block, a baton (per-block semaphores) serialises them ----------------
struct CoopGrid {
  int n = 0;
  std::vector<sem_t> sem;
  std::vector<char> live;
  bool reverse = false;
  unsigned long long yields = 0, yield_limit = 0;
};
static CoopGrid* g_coop = nullptr;                 // the co-resident launch in flight (launches are serialised by g_mu)
static thread_local int g_coop_me = -1;            // this thread's block number in it

static int coop_next(CoopGrid* g, int me) {        // next live block in the round-robin order, -1 if none but `me`
  for (int i = 1; i <= g->n; i++) {
    const int b = g->reverse ? ((me - i) % g->n + g->n) % g->n : (me + i) % g->n;
    if (b != me && g->live[b]) return b;
  }
  return -1;
}

void* dyn_shared() { return g_dyn.data(); }

static void yield_to_scheduler() {
  Fiber* me = g_cur;
  hipemu_switch(&me->ctx, &g_sched);
}
void syncthreads() {
  g_cur->state = AT_BARRIER;
  yield_to_scheduler();
}
void wave_op(int op, const void* in0, const void* in1, const void* in2, void* out, int i0, int i1, int i2, int i3) {
  Fiber* f = g_cur;
  f->op = op; f->in0 = in0; f->in1 = in1; f->in2 = in2; f->out = out;
  f->imm[0] = i0; f->imm[1] = i1; f->imm[2] = i2; f->imm[3] = i3;
  f->state = AT_WAVE;
  yield_to_scheduler();
}

void spin_yield() {
  g_cur->state = AT_SPIN;
  yield_to_scheduler();
}

static void trampoline() {
  (*g_body)();
  g_cur->state = DONE;
  hipemu_switch(&g_cur->ctx, &g_sched);
  abort();   // a finished fiber is never resumed
}

static void die(const char* msg) {
  fprintf(stderr, "hipemu: %s\n", msg);
  abort();
}

// the running block cannot go on until another block acts: hand the baton to the next live block and wait for it
static void coop_yield() {
  CoopGrid* g = g_coop;
  if (!g || g_coop_me < 0) die("a thread spins (s_sleep) in an ORDINARY launch: its blocks run one after another, so nothing can satisfy the wait -- launch co-resident");
  const int nx = coop_next(g, g_coop_me);
  if (nx < 0) die("co-resident launch: the last live block spins on a condition nobody can satisfy (deadlock)");
  if (++g->yields > g->yield_limit) die("co-resident launch: yield budget exhausted (HIPEMU_YIELD_LIMIT) -- deadlock or livelock");
  sem_post(&g->sem[nx]);
  sem_wait(&g->sem[g_coop_me]);
}

// ---- the collectives ------------------------------------------------------------------------------------------
static inline int ld_i(const void* p) { int v; memcpy(&v, p, 4); return v; }
static inline void st_i(void* p, int v) { memcpy(p, &v, 4); }

// source lane of a DPP control for destination lane L (-1: out of range)
static int dpp_source(int ctrl, int L) {
  const int row = L & ~15, r = L & 15;
  if (ctrl >= 0x000 && ctrl <= 0x0FF) return (L & ~3) | ((ctrl >> (2 * (L & 3))) & 3);
  if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl & 15; return r + n > 15 ? -1 : L + n; }
  if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl & 15; return r - n < 0 ? -1 : L - n; }
  if (ctrl >= 0x121 && ctrl <= 0x12F) { const int n = ctrl & 15; return row | ((r - n) & 15); }
  switch (ctrl) {
    case 0x130: return L + 1 > 63 ? -1 : L + 1;
    case 0x134: return (L + 1) & 63;
    case 0x138: return L - 1 < 0 ? -1 : L - 1;
    case 0x13C: return (L - 1) & 63;
    case 0x140: return row | (15 - r);
    case 0x141: return (L & ~7) | (7 - (L & 7));
    case 0x142: return row == 0 ? -1 : row - 1;          // lane 15 of the previous row
    case 0x143: return L < 32 ? -1 : 31;                 // lane 31 to rows 2 and 3
  }
  die("unsupported DPP control");
  return -1;
}

static void run_collective(Fiber** lane, int n_lanes) {
  // the participants: lanes parked at a wave collective; they must agree on the operation
  int op = 0;
  bool active[64];
  for (int l = 0; l < 64; l++) {
    active[l] = l < n_lanes && lane[l]->state == AT_WAVE;
    if (active[l]) {
      if (op == 0) op = lane[l]->op;
      else if (op != lane[l]->op) die("lanes of one wave reached DIFFERENT collectives (divergent wave operation)");
    }
  }
  switch (op) {
    case OP_WAVE_BARRIER: break;
    case OP_READFIRSTLANE: {
      int v = 0;
      for (int l = 0; l < 64; l++) if (active[l]) { v = ld_i(lane[l]->in0); break; }
      for (int l = 0; l < 64; l++) if (active[l]) st_i(lane[l]->out, v);
    } break;
    case OP_READLANE: {
      for (int l = 0; l < 64; l++) if (active[l]) {
        const int s = lane[l]->imm[0] & 63;
        if (s >= n_lanes) die("readlane of a lane beyond the block");
        // an inactive source lane still holds a register value on the GPU; here it has none
        if (!active[s]) die("readlane from a lane that is not at the collective");
        st_i(lane[l]->out, ld_i(lane[s]->in0));
      }
    } break;
    case OP_SHFL: case OP_SHFL_XOR: {
      for (int l = 0; l < 64; l++) if (active[l]) {
        const int w = lane[l]->imm[1];
        int s = op == OP_SHFL_XOR ? (l ^ lane[l]->imm[0]) : ((l & ~(w - 1)) | (lane[l]->imm[0] & (w - 1)));
        if (s < 0 || s >= 64 || !active[s]) s = l;
        st_i(lane[l]->out, ld_i(lane[s]->in0));
      }
    } break;
    case OP_BALLOT: {
      unsigned long long m = 0;
      for (int l = 0; l < 64; l++) if (active[l] && ld_i(lane[l]->in0)) m |= 1ull << l;
      for (int l = 0; l < 64; l++) if (active[l]) memcpy(lane[l]->out, &m, 8);
    } break;
    case OP_DPP: {
      int res[64];
      for (int l = 0; l < 64; l++) if (active[l]) {
        const Fiber* f = lane[l];
        const int ctrl = f->imm[0], row_mask = f->imm[1], bank_mask = f->imm[2], bound = f->imm[3];
        const int old = ld_i(f->in0);
        if (!((row_mask >> (l >> 4)) & 1) || !((bank_mask >> ((l >> 2) & 3)) & 1)) { res[l] = old; continue; }
        const int s = dpp_source(ctrl, l);
        if (s < 0 || !active[s]) res[l] = bound ? 0 : old;
        else res[l] = ld_i(lane[s]->in1);
      }
      for (int l = 0; l < 64; l++) if (active[l]) st_i(lane[l]->out, res[l]);
    } break;
    case OP_PERMLANE32_SWAP: case OP_PERMLANE16_SWAP: {
      // v_permlane32_swap vdst, vsrc: lanes [32, 64) of vdst trade places with lanes [0, 32) of vsrc;
      // v_permlane16_swap: the odd rows of vdst with the even rows of vsrc.  Returns {vdst', vsrc'}.
      const int h = op == OP_PERMLANE32_SWAP ? 32 : 16;
      unsigned vd[64], vs[64], nd[64], ns[64];
      for (int l = 0; l < 64; l++) {
        if (!active[l]) die("permlane swap with inactive lanes");
        vd[l] = (unsigned)ld_i(lane[l]->in0); vs[l] = (unsigned)ld_i(lane[l]->in1);
      }
      for (int l = 0; l < 64; l++) {
        if (l & h) { nd[l] = vs[l - h]; ns[l] = vs[l]; }
        else { nd[l] = vd[l]; ns[l] = vd[l + h]; }
      }
      for (int l = 0; l < 64; l++) { unsigned r[2] = {nd[l], ns[l]}; memcpy(lane[l]->out, r, 8); }
    } break;
    case OP_MFMA_F32_32X32X2: case OP_MFMA_BF16_32X32X16: {
      // A: lane l holds A[i = l % 32][k = kk * (l / 32) + t]; B: B[k][j = l % 32];
      // D: register v of lane l is D[i = 8 (v / 4) + 4 (l / 32) + v % 4][j = l % 32]
      const int kk = op == OP_MFMA_F32_32X32X2 ? 1 : 8;
      static float A[32][16], B[16][32];
      for (int l = 0; l < 64; l++) {
        if (!active[l]) die("MFMA with inactive lanes");
        const float* a = (const float*)lane[l]->in0; const float* b = (const float*)lane[l]->in1;
        for (int t = 0; t < kk; t++) { A[l & 31][kk * (l >> 5) + t] = a[t]; B[kk * (l >> 5) + t][l & 31] = b[t]; }
      }
      for (int l = 0; l < 64; l++) {
        const float* c = (const float*)lane[l]->in2; float* d = (float*)lane[l]->out;
        const int j = l & 31;
        for (int v = 0; v < 16; v++) {
          const int i = 8 * (v >> 2) + 4 * (l >> 5) + (v & 3);
          float acc = c[v];
          for (int k = 0; k < 2 * kk; k++) acc = fmaf(A[i][k], B[k][j], acc);
          d[v] = acc;
        }
      }
    } break;
    case OP_MFMA_F32_16X16X4: {
      // v_mfma_f32_16x16x4_f32: A: lane l holds A[i = l % 16][k = l / 16]; B: B[k = l / 16][j = l % 16];
      // D: register v of lane l is D[i = 4 (l / 16) + v][j = l % 16]
      float A[16][4], B[4][16];
      for (int l = 0; l < 64; l++) {
        if (!active[l]) die("MFMA with inactive lanes");
        A[l & 15][l >> 4] = *(const float*)lane[l]->in0;
        B[l >> 4][l & 15] = *(const float*)lane[l]->in1;
      }
      for (int l = 0; l < 64; l++) {
        const float* c = (const float*)lane[l]->in2; float* d = (float*)lane[l]->out;
        const int j = l & 15;
        for (int v = 0; v < 4; v++) {
          const int i = 4 * (l >> 4) + v;
          float acc = c[v];
          for (int k = 0; k < 4; k++) acc = fmaf(A[i][k], B[k][j], acc);
          d[v] = acc;
        }
      }
    } break;
    case OP_MFMA_F16_16X16X32: {
      // v_mfma_f32_16x16x32_f16: A: lane l holds A[i = l % 16][k = 8 (l / 16) + t], t = 0..7; B: B[k = 8 (l / 16) + t][j = l % 16];
      // D: register v of lane l is D[i = 4 (l / 16) + v][j = l % 16]
      static float A[16][32], B[32][16];
      for (int l = 0; l < 64; l++) {
        if (!active[l]) die("MFMA with inactive lanes");
        const float* a = (const float*)lane[l]->in0; const float* b = (const float*)lane[l]->in1;
        for (int t = 0; t < 8; t++) { A[l & 15][8 * (l >> 4) + t] = a[t]; B[8 * (l >> 4) + t][l & 15] = b[t]; }
      }
      for (int l = 0; l < 64; l++) {
        const float* c = (const float*)lane[l]->in2; float* d = (float*)lane[l]->out;
        const int j = l & 15;
        for (int v = 0; v < 4; v++) {
          const int i = 4 * (l >> 4) + v;
          float acc = c[v];
          for (int k = 0; k < 32; k++) acc = fmaf(A[i][k], B[k][j], acc);
          d[v] = acc;
        }
      }
    } break;
    default: die("unknown wave collective");
  }
  for (int l = 0; l < n_lanes && l < 64; l++) if (lane[l]->state == AT_WAVE) lane[l]->state = RUNNABLE;
}

// ---- one block --------------------------------------------------------------------------------------------------
static void run_block(const std::function<void()>& body, dim3 grid, dim3 block, dim3 bidx) {
  const int nt = (int)(block.x * block.y * block.z);
  while ((int)g_fibers.size() < nt) {
    Fiber* f = new Fiber();
    f->stack = (char*)mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (f->stack == MAP_FAILED) die("fiber stack allocation failed");
    g_fibers.push_back(f);
  }
  g_body = &body;
  for (int t = 0; t < nt; t++) {
    Fiber* f = g_fibers[t];
    f->tc.tidx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
    f->tc.bidx = bidx; f->tc.bdim = block; f->tc.gdim = grid;
    f->state = RUNNABLE;
    // initial frame: [mxcsr | x87 cw][r15 r14 r13 r12 rbx rbp][return address = trampoline][0]; the slot of the
    // return address is 16-byte aligned, so the trampoline starts with the stack alignment of a called function
    uint64_t* top = (uint64_t*)(((uintptr_t)f->stack + STACK_BYTES - 64) & ~(uintptr_t)15);
    top[1] = 0;
    top[0] = (uint64_t)(uintptr_t)&trampoline;
    for (int i = 1; i <= 6; i++) top[-i] = 0;
    top[-7] = 0x1F80ull | (0x037Full << 32);
    f->ctx.sp = &top[-7];
  }
  const int n_waves = (nt + 63) / 64;
  int live = nt;
  // HIPEMU_ORDER=reverse: waves and lanes are scheduled in descending order.  A kernel whose result depends on the
  // order in which the waves of a block (or the lanes of a wave) reach an exchange through LDS / global memory is
  // missing a barrier; the default ascending order and the reversed one must give the same output.
  static const bool reverse = []() { const char* e = getenv("HIPEMU_ORDER"); return e && e[0] == 'r'; }();
  while (live > 0) {
    bool progress = false;
    for (int wi = 0; wi < n_waves; wi++) {
      const int w = reverse ? n_waves - 1 - wi : wi;
      Fiber** lane = &g_fibers[w * 64];
      const int n_lanes = std::min(64, nt - w * 64);
      bool again = true;
      while (again) {
        again = false;
        bool parked = false;
        for (int li = 0; li < n_lanes; li++) {
          const int l = reverse ? n_lanes - 1 - li : li;
          Fiber* f = lane[l];
          if (f->state == RUNNABLE) {
            g_cur = f; g_tc = &f->tc;
            hipemu_switch(&g_sched, &f->ctx);
            progress = true;
            if (f->state == DONE) live--;
          }
          parked |= f->state == AT_WAVE;
        }
        // no lane of this wave can run any more: the ones parked at a collective are its participants
        if (parked) { run_collective(lane, n_lanes); again = true; progress = true; }
      }
    }
    // no wave can run: threads that spin on another block's progress give way first (a block barrier must not open
    // while a thread of the block is still on its way to it)
    bool spin = false;
    for (int t = 0; t < nt; t++) spin |= g_fibers[t]->state == AT_SPIN;
    if (spin) {
      coop_yield();
      for (int t = 0; t < nt; t++) if (g_fibers[t]->state == AT_SPIN) g_fibers[t]->state = RUNNABLE;
      continue;
    }
    // every wave is now at the block barrier or finished
    bool any = false;
    for (int t = 0; t < nt; t++) any |= g_fibers[t]->state == AT_BARRIER;
    if (any) {
      for (int t = 0; t < nt; t++) if (g_fibers[t]->state == AT_BARRIER) g_fibers[t]->state = RUNNABLE;
      progress = true;
    }
    if (!progress) die("deadlock in a block");
  }
  g_cur = nullptr; g_tc = nullptr;
}

// HIPEMU_SEGV_TRACE=1: print the native backtrace of a faulting kernel thread (fibers run on their own stacks, which
// Python's faulthandler cannot walk)
static void segv_handler(int sig, siginfo_t* si, void*) {
  static const char msg[] = "hipemu: fatal signal in emulated code, backtrace:\n";
  (void)!write(2, msg, sizeof(msg) - 1);
  if (g_cur) {
    char buf[160];
    int n = snprintf(buf, sizeof buf, "  block (%u,%u,%u) thread %u, fault address %p (si_code %d)\n", g_cur->tc.bidx.x,
                     g_cur->tc.bidx.y, g_cur->tc.bidx.z, g_cur->tc.tidx.x, si->si_addr, si->si_code);
    (void)!write(2, buf, n);
  }
  void* bt[48];
  int n = backtrace(bt, 48);
  backtrace_symbols_fd(bt, n, 2);
  _exit(139);
}
static void install_segv_trace() {
  static bool done = false;
  if (done) return;
  done = true;
  const char* e = getenv("HIPEMU_SEGV_TRACE");
  if (!e || !*e || *e == '0') return;
  static char alt[1 << 16];
  stack_t ss; ss.ss_sp = alt; ss.ss_size = sizeof alt; ss.ss_flags = 0;
  sigaltstack(&ss, nullptr);
  struct sigaction sa; memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = segv_handler; sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
  sigaction(SIGSEGV, &sa, nullptr); sigaction(SIGBUS, &sa, nullptr);
}

static double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

static void run_grid(const std::function<void()>& body, dim3 grid, dim3 block, size_t shmem) {
  if (g_dyn.size() < shmem + 16) g_dyn.resize(shmem + 16);
  static const bool reverse = []() { const char* e = getenv("HIPEMU_ORDER"); return e && e[0] == 'r'; }();
  const unsigned long long nb = (unsigned long long)grid.x * grid.y * grid.z;
  for (unsigned long long i = 0; i < nb; i++) {      // blocks too: no kernel may depend on the block order
    const unsigned long long b = reverse ? nb - 1 - i : i;
    run_block(body, grid, block, dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((unsigned long long)grid.x * grid.y))));
  }
}

// co-resident launch: every block on a thread of its own (own thread-local LDS, own fibers); the baton starts at block 0
struct CoopArg { const std::function<void()>* body; dim3 grid, block; size_t shmem; int b; CoopGrid* g; };
static void* coop_thread(void* p) {
  CoopArg* a = (CoopArg*)p;
  CoopGrid* g = a->g;
  g_coop_me = a->b;
  sem_wait(&g->sem[a->b]);                          // my first turn
  if (g_dyn.size() < a->shmem + 16) g_dyn.resize(a->shmem + 16);
  const unsigned long long b = (unsigned long long)a->b;
  run_block(*a->body, a->grid, a->block, dim3((unsigned)(b % a->grid.x), (unsigned)((b / a->grid.x) % a->grid.y),
                                              (unsigned)(b / ((unsigned long long)a->grid.x * a->grid.y))));
  for (Fiber* f : g_fibers) { munmap(f->stack, STACK_BYTES); delete f; }
  g_fibers.clear();
  g->live[a->b] = 0;
  const int nx = coop_next(g, a->b);
  if (nx >= 0) sem_post(&g->sem[nx]);               // the baton moves on; the last block to finish just leaves
  return nullptr;
}
static void run_grid_coop(const std::function<void()>& body, dim3 grid, dim3 block, size_t shmem) {
  const unsigned long long nb = (unsigned long long)grid.x * grid.y * grid.z;
  if (nb > 4096) die("co-resident launch with more than 4096 blocks");
  CoopGrid g;
  g.n = (int)nb; g.sem.resize(nb); g.live.assign(nb, 1);
  { const char* e = getenv("HIPEMU_ORDER"); g.reverse = e && e[0] == 'r'; }
  { const char* e = getenv("HIPEMU_YIELD_LIMIT"); g.yield_limit = e ? strtoull(e, nullptr, 10) : 200000000ull; }
  for (auto& s : g.sem) sem_init(&s, 0, 0);
  std::vector<CoopArg> args(nb);
  std::vector<pthread_t> th(nb);
  pthread_attr_t at; pthread_attr_init(&at); pthread_attr_setstacksize(&at, 1 << 20);
  g_coop = &g;
  for (unsigned long long b = 0; b < nb; b++) {
    args[b] = CoopArg{&body, grid, block, shmem, (int)b, &g};
    if (pthread_create(&th[b], &at, coop_thread, &args[b])) die("pthread_create failed");
  }
  pthread_attr_destroy(&at);
  sem_post(&g.sem[g.reverse ? nb - 1 : 0]);
  for (unsigned long long b = 0; b < nb; b++) pthread_join(th[b], nullptr);
  g_coop = nullptr;
  for (auto& s : g.sem) sem_destroy(&s);
}

void enqueue(hipStream_t st, std::function<void()> body, dim3 grid, dim3 block, size_t shmem, hipEvent_t e0, hipEvent_t e1,
             bool coresident) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  install_segv_trace();
  if (!st) st = &g_null_stream;
  if (block.x * block.y * block.z == 0 || grid.x * grid.y * grid.z == 0) return;
  auto node = [body, grid, block, shmem, coresident]() {
    if (coresident) run_grid_coop(body, grid, block, shmem); else run_grid(body, grid, block, shmem);
  };
  if (st->capturing) { st->graph->nodes.push_back(node); return; }
  if (e0) e0->t_ms = now_ms();
  node();
  if (e1) e1->t_ms = now_ms();
}

}  // namespace hipemu

// ---- host API -------------------------------------------------------------------------------------------------------
using hipemu::g_mu;
hipError_t hipSetDevice(int d) { return d >= 0 && d < 64 ? hipSuccess : hipErrorInvalidValue; }   // (every "device" is this host)
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : e == hipErrorOutOfMemory ? "out of memory" : "hipemu error"; }
// HIPEMU_GUARD=1 -- an address sanitizer for the kernels: every allocation ends flush (to 16 bytes) against an
// inaccessible page, so a read or write past its end faults at the offending instruction (HIPEMU_SEGV_TRACE=1 prints
// the kernel's backtrace, block and thread), and fresh memory is filled with 0xFF bytes (NaN as float, -1 as int)
// so that a read of never-written memory poisons the result instead of happening to see zeros.
namespace {
struct GuardAlloc { void* base; size_t len; };
std::mutex g_alloc_mu;
std::unordered_map<void*, GuardAlloc> g_guarded;
bool guard_mode() {
  static const bool on = []() { const char* e = getenv("HIPEMU_GUARD"); return e && *e && *e != '0'; }();
  return on;
}
}  // namespace
hipError_t hipMalloc(void** p, size_t n) {
  *p = nullptr;
  if (!guard_mode()) {
    if (posix_memalign(p, 256, n ? n : 256)) return hipErrorOutOfMemory;
    return hipSuccess;
  }
  const size_t page = 4096, need = (std::max<size_t>(n, 1) + 15) & ~(size_t)15;
  const size_t len = ((need + page - 1) / page) * page + page;
  char* base = (char*)mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (base == MAP_FAILED) return hipErrorOutOfMemory;
  mprotect(base + len - page, page, PROT_NONE);
  char* user = base + len - page - need;
  static const bool poison = []() { const char* e = getenv("HIPEMU_GUARD"); return e && *e == '1'; }();
  if (poison) memset(user, 0xFF, need);
  std::lock_guard<std::mutex> lk(g_alloc_mu);
  g_guarded[user] = GuardAlloc{base, len};
  *p = user;
  return hipSuccess;
}
hipError_t hipFree(void* p) {
  if (!p) return hipSuccess;
  if (!guard_mode()) { free(p); return hipSuccess; }
  std::lock_guard<std::mutex> lk(g_alloc_mu);
  auto it = g_guarded.find(p);
  if (it == g_guarded.end()) return hipErrorInvalidValue;
  munmap(it->second.base, it->second.len);      // use after free faults as well
  g_guarded.erase(it);
  return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void* p) { return hipFree(p); }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (st && st->capturing) { st->graph->nodes.push_back([d, s, n]() { if (n) memmove(d, s, n); }); return hipSuccess; }
  if (n) memmove(d, s, n);
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t st) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  auto f = [d, dp, s, sp, w, h]() { for (size_t r = 0; r < h; r++) memmove((char*)d + r * dp, (const char*)s + r * sp, w); };
  if (st && st->capturing) { st->graph->nodes.push_back(f); return hipSuccess; }
  f();
  return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (st && st->capturing) { st->graph->nodes.push_back([d, v, n]() { memset(d, v, n); }); return hipSuccess; }
  memset(d, v, n);
  return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t* s) { *s = new ihipStream_t(); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new ihipStream_t(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new ihipEvent_t(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new ihipEvent_t(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st) {
  if (st && st->capturing) return hipSuccess;
  e->t_ms = hipemu::now_ms();
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t_ms - a->t_ms); return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t st, hipStreamCaptureMode) {
  if (!st || st->capturing) return hipErrorInvalidValue;
  st->capturing = true; st->graph = new ihipGraph();
  return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t st, hipGraph_t* g) {
  if (!st || !st->capturing) return hipErrorInvalidValue;
  st->capturing = false; *g = st->graph; st->graph = nullptr;
  return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t* ge, hipGraph_t g, hipGraphNode_t*, char*, size_t) {
  *ge = new ihipGraphExec(); (*ge)->nodes = g->nodes;
  return hipSuccess;
}
hipError_t hipGraphLaunch(hipGraphExec_t ge, hipStream_t) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  for (auto& n : ge->nodes) n();
  return hipSuccess;
}
hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t g) { delete g; return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t attr, int) {
  if (attr == hipDeviceAttributeCooperativeLaunch) {        // HIPEMU_NO_COOP=1: a device / partition without cooperative launches
    const char* c = getenv("HIPEMU_NO_COOP");
    *v = (c && c[0] == '1') ? 0 : 1;
    return hipSuccess;
  }
  if (attr != hipDeviceAttributeMultiprocessorCount) return hipErrorInvalidValue;
  const char* e = getenv("HIPEMU_CUS");
  *v = e ? atoi(e) : 256;
  return hipSuccess;
}
hipError_t hipemu_occupancy(int* blocks_per_cu) {
  const char* e = getenv("HIPEMU_BLOCKS_PER_CU");
  *blocks_per_cu = e ? atoi(e) : 1;
  return hipSuccess;
}
