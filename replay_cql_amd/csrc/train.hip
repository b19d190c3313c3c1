// Training-step driver (a8): enqueues the whole CQL step on one stream, no host sync, no allocation.
// Split in two halves so that a data-parallel caller can all-reduce ctx->grads (RCCL over xGMI) in between.
#include <math.h>
#include <stdlib.h>
#include "common.h"
#include "qhead_internal.h"

static int g_concurrency = -1;   // -1: read CQL_CONCURRENCY on first use (default on)

// ---- schedule marks (tools/phase_timing.py): a handful of timing events at the joints of ONE step of a
// cqlrec_train_steps call -- cheap enough not to disturb the overlap they measure (unlike bracketing every kernel)
enum { MK_LOSS = 0, MK_DH, MK_DE, MK_CHAIN, MK_ADAM_IN, MK_ADAM_OUT, MK_PROLOGUE, MK_LSE, MK_NEXT_LOSS, MK_ENC_DX, MK_GATHER_BWD, MK_SORT_NEXT, MK_COUNT };
static hipEvent_t g_marks[MK_COUNT];
static int g_marks_on = 0;       // cqlrec_debug_marks_enable
static int g_mark_phase = 0;     // 1: backward of the marked step, 2: forward of the step after it
static void mark(int k, hipStream_t s) {
  if (g_marks_on && g_mark_phase) (void)hipEventRecord(g_marks[k], s);
}
extern "C" int cqlrec_debug_marks_enable(int32_t on) {
  if (on && !g_marks[0])
    for (int i = 0; i < MK_COUNT; ++i)
      if (hipEventCreate(&g_marks[i]) != hipSuccess) return CQLREC_ERR_HIP;
  g_marks_on = on ? 1 : 0;
  return CQLREC_OK;
}
// ms of each mark relative to MK_LOSS (device must be idle: synchronises); order as the enum above
extern "C" int cqlrec_debug_marks_read(float* ms_out) {
  if (!g_marks[0] || !ms_out) return CQLREC_ERR_INVALID;
  for (int i = 0; i < MK_COUNT; ++i) {
    ms_out[i] = -1.f;
    if (hipEventSynchronize(g_marks[i]) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_marks[MK_LOSS], g_marks[i]) == hipSuccess) ms_out[i] = ms;
  }
  return CQLREC_OK;
}

namespace {
struct Carve {
  char* base;
  int64_t off;
  explicit Carve(void* p) : base((char*)p), off(0) {}
  template <typename T>
  T* take(int64_t n) {
    T* p = base ? (T*)(base + off) : nullptr;
    off += (n * (int64_t)sizeof(T) + 255) / 256 * 256;
    return p;
  }
};

struct StepWs {
  int32_t *users, *tpos, *act, *a_star;
  float *rew, *done, *q_a, *lse, *nlse2, *nlse_nat, *q_targ, *y, *coef, *maxv, *loss, *term;
  float *h0_s, *dH, *dh0;
  uint16_t *h0b, *zb, *hb, *h0b_t, *zb_t, *hb_t;
  void *ws_q, *ws_q2, *ws_qb, *ws_qb2, *ws_enc, *ws_gb, *ws_oh;
  int64_t ws_q_bytes, ws_qb_bytes, ws_qf_bytes, ws_enc_bytes, ws_gb_bytes, ws_oh_bytes;
  int64_t total;
};

// The per-step vectors and activations exist twice (parity = step & 1): cqlrec_train_steps starts the prologue of step
// t+1 (sample, gathers, encoder) while the item-side backward of step t still reads act / coef / nlse2 / hb of step t.
void carve_small(Carve& c, StepWs& w, int32_t B, int32_t d) {
  w.users = c.take<int32_t>(B);
  w.tpos = c.take<int32_t>(B);
  w.act = c.take<int32_t>(B);
  w.a_star = c.take<int32_t>(B);
  w.rew = c.take<float>(B);
  w.done = c.take<float>(B);
  w.q_a = c.take<float>(B);
  w.lse = c.take<float>(B);
  w.nlse2 = c.take<float>(B);
  w.nlse_nat = c.take<float>(B);
  w.q_targ = c.take<float>(B);
  w.y = c.take<float>(B);
  w.coef = c.take<float>(B);
  w.maxv = c.take<float>(B);
  w.loss = c.take<float>(64);
  w.term = c.take<float>(B);      // per-transition loss terms (cql_td_coef -> cql_td_loss_sum)
  w.h0_s = c.take<float>((int64_t)B * d);
  w.dH = c.take<float>((int64_t)B * d);
  w.dh0 = c.take<float>((int64_t)B * d);
  w.h0b = c.take<uint16_t>((int64_t)2 * B * d);
  w.zb = c.take<uint16_t>((int64_t)2 * B * d);
  w.hb = c.take<uint16_t>((int64_t)2 * B * d);
  w.h0b_t = c.take<uint16_t>((int64_t)B * d);
  w.zb_t = c.take<uint16_t>((int64_t)B * d);
  w.hb_t = c.take<uint16_t>((int64_t)B * d);
}

StepWs carve_step(void* ws, int32_t B, int64_t N, int32_t d, int32_t L, uint64_t step = 0) {
  StepWs w, other;
  Carve c(ws);
  if (step & 1) {
    carve_small(c, other, B, d);
    carve_small(c, w, B, d);
  } else {
    carve_small(c, w, B, d);
    carve_small(c, other, B, d);
  }
  w.ws_q_bytes = cqlrec_qhead_ws_bytes(B, N, d);
  w.ws_q = c.take<char>(w.ws_q_bytes);     // LSE partials (branch A)
  w.ws_q2 = c.take<char>(w.ws_q_bytes);    // ARGMAX partials (branch B runs concurrently)
  w.ws_qb_bytes = cqlrec_qhead_bwd_ws_bytes(B, N, d);
  w.ws_qf_bytes = w.ws_qb_bytes + w.ws_q_bytes;
  w.ws_qb = c.take<char>(w.ws_qf_bytes);    // fused forward of branch A: slabs of the soft part of dH + LSE partials
  w.ws_qb2 = c.take<char>(w.ws_qb_bytes);   // item-side backward: its own scratch, the two kernels run concurrently
  w.ws_enc_bytes = cqlrec_encoder_bwd_ws_bytes(B, d);
  w.ws_enc = c.take<char>(w.ws_enc_bytes);
  w.ws_gb_bytes = cqlrec_gather_pool_bwd_ws_bytes(B, L, d);
  void* gb0 = c.take<char>(w.ws_gb_bytes);   // sorted (item, state) pairs: by parity as well, the sort of step t+1 is
  void* gb1 = c.take<char>(w.ws_gb_bytes);   // issued while the gather backward of step t has not run yet
  w.ws_gb = (step & 1) ? gb1 : gb0;
  w.ws_oh_bytes = cql_onehot_ws_bytes(B, d);
  void* oh0 = c.take<char>(w.ws_oh_bytes);   // sorted (action, transition) pairs of the one-hot scatter, by parity too
  void* oh1 = c.take<char>(w.ws_oh_bytes);
  w.ws_oh = (step & 1) ? oh1 : oh0;
  w.total = c.off;
  return w;
}

int check_ctx(const cqlrec_train_ctx* c) {
  CQL_REQUIRE(c != nullptr, "train_step: ctx is NULL");
  CQL_REQUIRE(c->offsets && c->items && c->rewards, "train_step: CSR pointers are NULL");
  CQL_REQUIRE(c->theta && c->grads && c->adam_m && c->adam_v && c->target && c->theta_b && c->target_b,
              "train_step: model state pointers are NULL");
  CQL_REQUIRE(c->ws != nullptr, "train_step: workspace is NULL");
  CQL_REQUIRE(c->batch > 0 && c->window > 0 && c->world > 0 && c->rank >= 0 && c->rank < c->world,
              "train_step: batch=%d window=%d world=%d rank=%d", c->batch, c->window, c->world, c->rank);
  const int32_t d = c->layout.d;
  CQL_REQUIRE(d == 64 || d == 128 || d == 256, "train_step: d=%d unsupported", d);
  CQL_REQUIRE(c->ws_bytes >= cqlrec_train_ws_bytes(c->batch, c->layout.n_items, d, c->window),
              "train_step: workspace too small");
  return CQLREC_OK;
}
}  // namespace

// 1 (default): independent parts of the step run on internal side streams (fork/join by events); 0: everything on the
// caller's stream in program order (used for per-kernel timing, where overlapped launches would blur the durations)
extern "C" int cqlrec_set_concurrency(int32_t on) {
  g_concurrency = on ? 1 : 0;
  return CQLREC_OK;
}

extern "C" int64_t cqlrec_train_ws_bytes(int32_t batch, int64_t n_items, int32_t d, int32_t window) {
  return carve_step(nullptr, batch, n_items, d, window).total + 256;
}

#define CQL_TRY(expr)            \
  do {                           \
    int rc__ = (expr);           \
    if (rc__ != CQLREC_OK) return rc__; \
  } while (0)

namespace {
// side stream for the part of the gather backward that depends only on the sampled batch (pairs + radix sort): it
// runs underneath the Q-head kernels.  Fork/join through events, so the structure stays capturable in a hipGraph.
struct SideStream {
  hipStream_t s = nullptr;    // sort of the gather backward (forward phase), item-side backward (backward phase)
  hipStream_t s2 = nullptr;   // branch B of the forward (s' rows: argmax + target network)
  hipStream_t s3 = nullptr;   // sampling + sorts of the NEXT step (cqlrec_train_steps)
  hipEvent_t sorted[2] = {nullptr, nullptr};   // sorted pairs of the step with this parity are in place
  hipEvent_t forked = nullptr, fork2 = nullptr, join2 = nullptr;
  hipEvent_t loss = nullptr, items = nullptr, dh = nullptr, eout = nullptr, presample = nullptr, adam_in = nullptr, bpro = nullptr, fwd_done = nullptr;
  bool ok = false;
  bool tried = false;
};
bool concurrency_on() {
  if (g_concurrency < 0) {
    const char* v = getenv("CQL_CONCURRENCY");
    g_concurrency = (v && *v == '0') ? 0 : 1;
  }
  return g_concurrency != 0;
}
// ---- which streams may run beside each other -----------------------------------------------------------------------------
// Measured on MI355X / ROCm 7.2 (tools/probes/pipe_probe.hip, profiles/r03_pipe_probe.txt): a process's hardware queues sit
// on the 4 compute pipes round-robin in the order in which its streams were first used, and while a grid larger than the
// chip is being handed out on one queue, NO kernel of another queue on the same pipe is dispatched.  The step driver runs
// four streams side by side (the caller's + three); whether two of them share a pipe used to depend on what else the
// process had created streams for, and in which order -- a torch.cuda.Stream() between the log generation and the first
// step, or the model constructed before the first kernel on the default stream, cost the step ALL of its concurrency
// (cfg3: 1.33 instead of 0.69 ms; tools/stream_order_probe3.py).  So the streams are CHOSEN: out of a handful of fresh
// candidates, three that block neither the caller's (default) stream nor each other, by the same two-kernel test.
__global__ __launch_bounds__(64) void cql_pipe_hog_kernel(int spins) {
  extern __shared__ volatile char pad[];      // 60 KiB of dynamic LDS: two blocks per CU resident, the rest of the grid waits
  pad[threadIdx.x] = 1;
  for (int i = 0; i < spins; ++i) __builtin_amdgcn_s_sleep(127);
  pad[threadIdx.x + 64] = pad[threadIdx.x];
}
__global__ void cql_pipe_tiny_kernel() {}

struct PipeProbe {
  hipEvent_t a0 = nullptr, a1 = nullptr, b1 = nullptr;
  bool ok = false;
  PipeProbe() {
    ok = hipEventCreate(&a0) == hipSuccess && hipEventCreate(&a1) == hipSuccess && hipEventCreate(&b1) == hipSuccess;
  }
  ~PipeProbe() {
    if (a0) (void)hipEventDestroy(a0);
    if (a1) (void)hipEventDestroy(a1);
    if (b1) (void)hipEventDestroy(b1);
  }
  // 1: a kernel on y is not dispatched while a large grid on x is being handed out; 0: it is; -1: the test failed
  int conflict(hipStream_t x, hipStream_t y) {
    if (!ok || hipDeviceSynchronize() != hipSuccess) return -1;
    (void)hipEventRecord(a0, x);
    hipLaunchKernelGGL(cql_pipe_hog_kernel, dim3(8192), dim3(64), 60 * 1024, x, 3);     // ~0.2 ms
    (void)hipEventRecord(a1, x);
    hipLaunchKernelGGL(cql_pipe_tiny_kernel, dim3(1), dim3(64), 0, y);
    (void)hipEventRecord(b1, y);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return -1;
    float t_hog = 0.f, t_tiny = 0.f;
    if (hipEventElapsedTime(&t_hog, a0, a1) != hipSuccess || hipEventElapsedTime(&t_tiny, a0, b1) != hipSuccess) return -1;
    return (t_hog > 0.02f && t_tiny > 0.5f * t_hog) ? 1 : 0;
  }
};

hipStream_t g_aux[CQL_MAX_DEVICES][CQLREC_AUX_STREAMS] = {};
int g_pick_report[CQL_MAX_DEVICES] = {};     // candidates looked at (0: no probe ran), for cqlrec_runtime_report

// three streams for the step driver + one more for the caller, all first used HERE, in this order
bool pick_streams(hipStream_t main, hipStream_t out[3], hipStream_t* extra, int* looked_at) {
  constexpr int NC = 8;
  static const bool probe_on = !(getenv("CQL_PIPE_PROBE") && getenv("CQL_PIPE_PROBE")[0] == '0');
  hipStream_t cand[NC] = {};
  bool used[NC] = {};
  int n_out = 0;
  *looked_at = 0;
  hipLaunchKernelGGL(cql_pipe_tiny_kernel, dim3(1), dim3(64), 0, main);       // the caller's queue exists first
  PipeProbe pp;
  int made = 0;
  for (; made < NC && n_out < 3; ++made) {
    if (hipStreamCreateWithFlags(&cand[made], hipStreamNonBlocking) != hipSuccess) break;
    hipLaunchKernelGGL(cql_pipe_tiny_kernel, dim3(1), dim3(64), 0, cand[made]);   // binds its queue now: creation order
    bool good = true;
    if (probe_on && pp.ok) {
      *looked_at = made + 1;
      good = pp.conflict(main, cand[made]) == 0;
      for (int k = 0; good && k < n_out; ++k) good = pp.conflict(out[k], cand[made]) == 0;
    }
    if (good) {
      out[n_out++] = cand[made];
      used[made] = true;
    }
  }
  // not enough independent ones (queues exhausted, probe failed): take what there is, as before
  for (int i = 0; i < made && n_out < 3; ++i)
    if (!used[i]) {
      out[n_out++] = cand[i];
      used[i] = true;
    }
  while (n_out < 3) {
    hipStream_t st;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return false;
    out[n_out++] = st;
  }
  // the caller's extra stream: an unused candidate that does not block the caller's main stream, else a fresh one
  *extra = nullptr;
  for (int i = 0; i < made; ++i)
    if (!used[i] && !*extra && (!probe_on || !pp.ok || pp.conflict(main, cand[i]) == 0)) {
      *extra = cand[i];
      used[i] = true;
    }
  for (int i = 0; i < made; ++i)
    if (!used[i]) (void)hipStreamDestroy(cand[i]);
  if (!*extra && hipStreamCreateWithFlags(extra, hipStreamNonBlocking) != hipSuccess) return false;
  return hipDeviceSynchronize() == hipSuccess;
}

SideStream& side_stream() {
  static SideStream per_device[CQL_MAX_DEVICES];     // streams and events belong to the device they were created on
  static SideStream off;   // never ok: serial mode
  if (!concurrency_on()) return off;
  const int slot = cql_device_slot();
  SideStream& ss = per_device[slot];
  if (!ss.tried) {
    ss.tried = true;
    // Plain streams, all of one priority class (mixing priority classes did not steer the dispatcher).  s2 = forward branch,
    // s = sorts of the forward / item side of the single-rank drivers, s3 = sample-ahead of cqlrec_train_steps.  A
    // data-parallel caller never runs cqlrec_train_steps: ITS item-side stream (cqlrec_aux_stream(0)) is s3.
    hipStream_t pk[3] = {};
    hipStream_t extra = nullptr;
    ss.ok = pick_streams(nullptr, pk, &extra, &g_pick_report[slot]);
    if (ss.ok) {
      ss.s2 = pk[0];
      ss.s = pk[1];
      ss.s3 = pk[2];
      if (!g_aux[slot][0]) g_aux[slot][0] = pk[2];      // (already set: runtime_init ran while the library was in serial mode)
      if (!g_aux[slot][1]) g_aux[slot][1] = extra;
    }
    ss.ok = ss.ok &&
            hipEventCreateWithFlags(&ss.sorted[0], hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ss.sorted[1], hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ss.forked, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ss.fork2, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ss.join2, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ss.loss, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ss.items, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ss.dh, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ss.eout, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ss.presample, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ss.adam_in, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ss.bpro, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ss.fwd_done, hipEventDisableTiming) == hipSuccess;
  }
  return ss;
}

// (the three streams exist as soon as side_stream() has run)
bool need_side_streams(SideStream& ss, bool want_s3 = false) {
  (void)want_s3;
  return ss.ok;
}

struct StepPtrs {
  const uint16_t *Ein_b, *Eout_b, *W1_b, *W2_b, *tEin_b, *tEout_b, *tW1_b, *tW2_b;
  const float *b_out, *b1, *b2, *tb_out, *tb1, *tb2;
};
// alpha / (global batch), formed in double and rounded once (the oracle's np.float32(alpha / Bg))
float alpha_scale(const cqlrec_train_ctx* c) { return (float)(c->alpha / ((double)c->batch * (double)c->world)); }
StepPtrs step_ptrs(const cqlrec_train_ctx* c) {
  const cqlrec_layout& L = c->layout;
  StepPtrs p;
  p.Ein_b = c->theta_b + L.off_E_in;   p.Eout_b = c->theta_b + L.off_E_out;
  p.W1_b = c->theta_b + L.off_W1;      p.W2_b = c->theta_b + L.off_W2;
  p.b_out = c->theta + L.off_b_out;    p.b1 = c->theta + L.off_b1;   p.b2 = c->theta + L.off_b2;
  p.tEin_b = c->target_b + L.off_E_in; p.tEout_b = c->target_b + L.off_E_out;
  p.tW1_b = c->target_b + L.off_W1;    p.tW2_b = c->target_b + L.off_W2;
  p.tb_out = c->target + L.off_b_out;  p.tb1 = c->target + L.off_b1; p.tb2 = c->target + L.off_b2;
  return p;
}
}  // namespace

extern "C" int cqlrec_runtime_init(void) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) {
    cql_set_error("runtime_init: no HIP device");
    return CQLREC_ERR_HIP;
  }
  const int slot = cql_device_slot();
  if (concurrency_on()) {
    if (!side_stream().ok) {
      cql_set_error("runtime_init: creating the internal streams failed");
      return CQLREC_ERR_HIP;
    }
  } else {      // strict program order inside the library: the caller's side streams are plain ones
    for (int i = 0; i < CQLREC_AUX_STREAMS; ++i)
      if (!g_aux[slot][i] && hipStreamCreateWithFlags(&g_aux[slot][i], hipStreamNonBlocking) != hipSuccess) {
        cql_set_error("runtime_init: creating aux stream %d failed", i);
        return CQLREC_ERR_HIP;
      }
  }
  return CQLREC_OK;
}

extern "C" int cqlrec_runtime_probe_count(void) { return g_pick_report[cql_device_slot()]; }

extern "C" cqlrec_stream cqlrec_aux_stream(int32_t index) {
  if (index < 0 || index >= CQLREC_AUX_STREAMS) return nullptr;
  return (cqlrec_stream)g_aux[cql_device_slot()][index];
}


#define CQL_HIP_TRY(expr, what)                 \
  do {                                          \
    if ((expr) != hipSuccess) {                 \
      cql_set_error("%s: %s failed", what, #expr); \
      return CQLREC_ERR_HIP;                    \
    }                                           \
  } while (0)

namespace {
// sample + forward + loss (+ the sort for the gather backward, forked onto the side stream).  `eout_ready`: event
// after which the item-side parameters (E_out, b_out and their shadows) are up to date -- everything before the
// Q-head kernels (sample, window gathers, encoder) reads only E_in / W1 / W2 and may start earlier.
// `presampled`: event after which this step's transitions are already in place (sample_ahead); the sorted pairs the
// backward needs follow under ss.sorted[step & 1].
int sample_ahead(const cqlrec_train_ctx* c, uint64_t step, cqlrec_stream stream, hipEvent_t sampled_ev) {
  const cqlrec_layout& L = c->layout;
  StepWs w = carve_step(c->ws, c->batch, L.n_items, L.d, c->window, step);
  CQL_TRY(cqlrec_sample_transitions(c->offsets, c->items, c->rewards, c->n_users, c->seed, step,
                                    (uint64_t)c->rank * (uint64_t)c->batch, c->batch, w.users, w.tpos, w.act, w.rew,
                                    w.done, stream));
  CQL_HIP_TRY(hipEventRecord(sampled_ev, (hipStream_t)stream), "train_steps");
  CQL_TRY(cqlrec_gather_pool_bwd_prepare(c->offsets, c->items, w.users, w.tpos, 0, c->batch, c->window, L.d, L.n_items,
                                         w.ws_gb, w.ws_gb_bytes, stream));
  return cql_onehot_prepare(w.act, c->batch, L.n_items, L.d, w.ws_oh, w.ws_oh_bytes, (hipStream_t)stream);
}

int backward_items_long_impl(const cqlrec_train_ctx* c, uint64_t step, cqlrec_stream stream, CqlAdamFix* fix);
static bool dh_early() {
  static const bool v = !(getenv("CQL_DH_EARLY") && getenv("CQL_DH_EARLY")[0] == '0');
  return v;
}
// early_items (a stream) + early_fix: the long item-side kernel of THIS step is launched on that stream as soon as the fused
// forward has produced -lse (it needs nothing from the loss), see backward_items_impl
// the loss VALUE of `step` from the terms its forward left (nothing in the step waits for it)
int loss_sum_impl(const cqlrec_train_ctx* c, uint64_t step, float* loss_out, hipStream_t on) {
  StepWs w = carve_step(c->ws, c->batch, c->layout.n_items, c->layout.d, c->window, step);
  const float inv_batch = 1.0f / ((float)c->batch * (float)c->world);
  return cql_td_loss_sum(w.term, c->batch, inv_batch, loss_out ? loss_out : w.loss, on);
}
// loss_sum_deferred != NULL: the loss terms are left for the caller's loss_sum_impl (on a stream that has slack)
int forward_impl(const cqlrec_train_ctx* c, uint64_t step, float* loss_out, cqlrec_stream stream, hipEvent_t eout_ready,
                 hipEvent_t presampled = nullptr, hipStream_t early_items = nullptr, CqlAdamFix* early_fix = nullptr,
                 bool* loss_sum_deferred = nullptr) {
  const cqlrec_layout& L = c->layout;
  const int32_t B = c->batch, d = L.d, W = c->window;
  const int64_t N = L.n_items;
  StepWs w = carve_step(c->ws, B, N, d, W, step);
  const StepPtrs p = step_ptrs(c);
  const int64_t Bd = (int64_t)B * d;
  hipStream_t s = (hipStream_t)stream;

  SideStream& ss = side_stream();
  if (presampled) {
    CQL_HIP_TRY(hipStreamWaitEvent(s, presampled, 0), "train_step_forward");
  } else {
    // transitions of this rank's slots of the global step
    CQL_TRY(cqlrec_sample_transitions(c->offsets, c->items, c->rewards, c->n_users, c->seed, step,
                                      (uint64_t)c->rank * (uint64_t)B, B, w.users, w.tpos, w.act, w.rew, w.done, stream));
    if (ss.ok && need_side_streams(ss)) {
      if (hipEventRecord(ss.forked, s) != hipSuccess || hipStreamWaitEvent(ss.s, ss.forked, 0) != hipSuccess) ss.ok = false;
    }
  }
  if (presampled) {
  } else if (ss.ok) {
    CQL_TRY(cqlrec_gather_pool_bwd_prepare(c->offsets, c->items, w.users, w.tpos, 0, B, W, d, N, w.ws_gb, w.ws_gb_bytes,
                                           (cqlrec_stream)ss.s));
    CQL_TRY(cql_onehot_prepare(w.act, B, N, d, w.ws_oh, w.ws_oh_bytes, ss.s));
    CQL_HIP_TRY(hipEventRecord(ss.sorted[step & 1], ss.s), "train_step_forward");
  } else {
    CQL_TRY(cqlrec_gather_pool_bwd_prepare(c->offsets, c->items, w.users, w.tpos, 0, B, W, d, N, w.ws_gb, w.ws_gb_bytes,
                                           stream));
    CQL_TRY(cql_onehot_prepare(w.act, B, N, d, w.ws_oh, w.ws_oh_bytes, s));
  }
  // The forward has two independent branches that meet at the TD target:
  //   A (this stream):  s  under theta  -> encoder -> logsumexp over the catalogue, Q(s, a)
  //   B (side stream):  s' under theta  -> encoder -> argmax;  s' under the target net -> encoder -> Q_target(s', a*)
  // Run concurrently, the small gather / encoder / finalize launches of one branch fill the launch gaps of the other.
  cqlrec_stream sb = stream;
  const bool par = ss.ok && hipEventRecord(ss.fork2, s) == hipSuccess && hipStreamWaitEvent(ss.s2, ss.fork2, 0) == hipSuccess;
  if (par) sb = (cqlrec_stream)ss.s2;
  // CQL_ARGMAX_CORUN=1 (d = 128; A/B knob, default off): the two catalogue passes are launched TOGETHER -- the ARGMAX pass
  // in its small-wave form (cql_qhead_argmax_beside), whose waves fit beside the fused forward's on a SIMD -- with the
  // prologue of branch B enqueued first and branch A waiting for it in front of its Q-head pass.  Measured: the step
  // gets 3.5 % SLOWER (0.783 vs 0.757 ms): the fused forward is bound by instruction issue, not by the MFMA pipe, and a
  // second wave on the SIMD takes issue slots from it.
  static const bool corun_on = getenv("CQL_ARGMAX_CORUN") && getenv("CQL_ARGMAX_CORUN")[0] == '1';
  const bool corun = par && d == 128 && corun_on;
  auto branch_b_prologue = [&]() -> int {
    CQL_TRY(cqlrec_gather_pool_fwd(p.Ein_b, c->offsets, c->items, w.users, w.tpos, 1, B, W, d, nullptr, w.h0b + Bd, nullptr, sb));
    CQL_TRY(cqlrec_gather_pool_fwd(p.tEin_b, c->offsets, c->items, w.users, w.tpos, 1, B, W, d, nullptr, w.h0b_t, nullptr, sb));
    CQL_TRY(cqlrec_encoder_fwd(w.h0b + Bd, p.W1_b, p.b1, p.W2_b, p.b2, B, d, w.zb + Bd, w.hb + Bd, sb));
    CQL_TRY(cqlrec_encoder_fwd(w.h0b_t, p.tW1_b, p.tb1, p.tW2_b, p.tb2, B, d, w.zb_t, w.hb_t, sb));
    return CQLREC_OK;
  };
  if (corun) {
    CQL_TRY(branch_b_prologue());
    CQL_HIP_TRY(hipEventRecord(ss.bpro, ss.s2), "train_step_forward");
  }
  // ---- branch A
  CQL_TRY(cqlrec_gather_pool_fwd(p.Ein_b, c->offsets, c->items, w.users, w.tpos, 0, B, W, d, w.h0_s, w.h0b, nullptr, stream));
  CQL_TRY(cqlrec_encoder_fwd(w.h0b, p.W1_b, p.b1, p.W2_b, p.b2, B, d, w.zb, w.hb, stream));
  if (g_mark_phase == 2) mark(MK_PROLOGUE, s);
  CQL_TRY(cql_qhead_fwd_lse_dh_prepare(w.ws_qb, B, N, d, s));      // (a 4-byte memset: in front of the wait, not behind it)
  if (eout_ready) CQL_HIP_TRY(hipStreamWaitEvent(s, eout_ready, 0), "train_step_forward");
  if (corun) CQL_HIP_TRY(hipStreamWaitEvent(s, ss.bpro, 0), "train_step_forward");
  // logsumexp AND the softmax-weighted sum of item rows (the soft part of dH) in ONE pass over the catalogue
  CQL_TRY(cql_qhead_fwd_lse_dh(w.hb, B, p.Eout_b, p.b_out, N, d, w.ws_qb, w.ws_qf_bytes, w.lse, w.nlse2, s, w.nlse_nat, 1));
  if (g_mark_phase == 2) mark(MK_LSE, s);
  if (early_items) {
    CQL_HIP_TRY(hipEventRecord(ss.fwd_done, s), "train_step_forward");
    CQL_HIP_TRY(hipStreamWaitEvent(early_items, ss.fwd_done, 0), "train_step_forward");
    CQL_TRY(backward_items_long_impl(c, step, (cqlrec_stream)early_items, early_fix));
  }
  // the part of dH that needs nothing from the loss, off the chain the next prologue waits for (CQL_DH_EARLY=0: A/B)
  if (dh_early())
    CQL_TRY(cql_qhead_dh_finish(w.ws_qb, B, N, d, w.lse, nullptr, w.act, p.Eout_b, alpha_scale(c), w.dH, s, 1));
  CQL_TRY(cqlrec_gather_dot(w.hb, p.Eout_b, p.b_out, w.act, B, d, w.q_a, stream));
  // ---- branch B
  if (!corun) CQL_TRY(branch_b_prologue());
  if (eout_ready && par) CQL_HIP_TRY(hipStreamWaitEvent(ss.s2, eout_ready, 0), "train_step_forward");
  if (corun)
    CQL_TRY(cql_qhead_argmax_beside(w.hb + Bd, B, p.Eout_b, p.b_out, N, d, w.ws_q2, w.ws_q_bytes, w.maxv, w.a_star,
                                    (hipStream_t)sb));
  else
    CQL_TRY(cql_qhead_argmax_step(w.hb + Bd, B, p.Eout_b, p.b_out, N, d, w.ws_q2, w.ws_q_bytes, w.maxv, w.a_star, (hipStream_t)sb));
  CQL_TRY(cqlrec_gather_dot(w.hb_t, p.tEout_b, p.tb_out, w.a_star, B, d, w.q_targ, sb));
  if (par && (hipEventRecord(ss.join2, ss.s2) != hipSuccess || hipStreamWaitEvent(s, ss.join2, 0) != hipSuccess)) {
    cql_set_error("train_step_forward: joining the forward branches failed");
    return CQLREC_ERR_HIP;
  }
  // loss + dQ coefficients
  const float inv_batch = 1.0f / ((float)B * (float)c->world);
  CQL_TRY(cql_td_coef(w.q_a, w.lse, w.q_targ, w.rew, w.done, B, (float)c->gamma, (float)c->alpha, inv_batch, w.coef, w.y,
                      w.term, s));
  if (loss_sum_deferred) *loss_sum_deferred = true;
  else CQL_TRY(cql_td_loss_sum(w.term, B, inv_batch, loss_out ? loss_out : w.loss, s));
  if (g_mark_phase == 2) mark(MK_NEXT_LOSS, s);
  return CQLREC_OK;
}

// item-side backward.  ctx->grads is zero on entry (contract of the step).  Three parts, the sum of a gradient row formed in
// ONE order on every path (pipelined, phased, strict) so that they stay bit-identical:
//   row = ((scale * dE piece holding stage 0) + one-hot part) + scale * cut pieces, in block order
// (1) the long kernel WRITES its rows (nothing to read, nothing to wait for but the forward's -lse: the pipelined driver
//     launches it as soon as the fused forward has finished, under the rest of the forward and the loss),
// (2) the one-hot part -- the only part that needs the loss's coefficients -- is added as a segmented sum,
// (3) the cut pieces: fix-up launch here, or left to the item-side optimizer (CqlAdamFix).
static bool onehot_atomic() {
  static const bool v = getenv("CQL_ONEHOT_ATOMIC") && getenv("CQL_ONEHOT_ATOMIC")[0] == '1';   // A/B knob: old order, float atomics
  return v;
}
int backward_items_long_impl(const cqlrec_train_ctx* c, uint64_t step, cqlrec_stream stream, CqlAdamFix* fix) {
  const cqlrec_layout& L = c->layout;
  const int32_t B = c->batch, d = L.d;
  StepWs w = carve_step(c->ws, B, L.n_items, d, c->window, step);
  const StepPtrs p = step_ptrs(c);
  return cql_qhead_bwd_items_long(w.hb, w.nlse2, w.coef, w.act, B, p.Eout_b, p.b_out, L.n_items, d, alpha_scale(c), w.ws_qb2,
                                  w.ws_qb_bytes, c->grads + L.off_E_out, c->grads + L.off_b_out, (hipStream_t)stream, fix,
                                  w.nlse_nat);
}
int backward_items_onehot_impl(const cqlrec_train_ctx* c, uint64_t step, cqlrec_stream stream) {
  const cqlrec_layout& L = c->layout;
  const int32_t B = c->batch, d = L.d;
  StepWs w = carve_step(c->ws, B, L.n_items, d, c->window, step);
  SideStream& ss = side_stream();
  // the (action, transition) pairs were sorted ahead of time on another stream (same event as the window pairs)
  if (ss.ok) CQL_HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, ss.sorted[step & 1], 0), "train_step_backward_items");
  return cql_onehot_apply(w.coef, w.hb, B, L.n_items, d, w.ws_oh, c->grads + L.off_E_out, c->grads + L.off_b_out,
                          (hipStream_t)stream, 1);
}
// all three in program order; defer: leave (3) to the caller's optimizer launch
int backward_items_impl(const cqlrec_train_ctx* c, uint64_t step, cqlrec_stream stream, CqlAdamFix* defer = nullptr) {
  const cqlrec_layout& L = c->layout;
  if (onehot_atomic()) {
    const int32_t B = c->batch, d = L.d;
    StepWs w = carve_step(c->ws, B, L.n_items, d, c->window, step);
    const StepPtrs p = step_ptrs(c);
    return cql_qhead_bwd_items_acc(w.hb, w.nlse2, w.coef, w.act, B, p.Eout_b, p.b_out, L.n_items, d, alpha_scale(c),
                                   w.ws_qb2, w.ws_qb_bytes, c->grads + L.off_E_out, c->grads + L.off_b_out,
                                   (hipStream_t)stream, 1, 0, -1, defer, w.nlse_nat);
  }
  CqlAdamFix fix = {};
  CQL_TRY(backward_items_long_impl(c, step, stream, &fix));
  CQL_TRY(backward_items_onehot_impl(c, step, stream));
  if (defer) *defer = fix;
  else CQL_TRY(cql_qde_fixup_deferred(fix, c->grads + L.off_E_out, c->grads + L.off_b_out, (hipStream_t)stream));
  return CQLREC_OK;
}

// records `dh` behind dh_finish, the last reader of the E_out shadow on this stream: the item-side Adam waits for it
int backward_states_impl(const cqlrec_train_ctx* c, uint64_t step, cqlrec_stream stream) {
  const cqlrec_layout& L = c->layout;
  const int32_t B = c->batch, d = L.d;
  const int64_t N = L.n_items;
  StepWs w = carve_step(c->ws, B, N, d, c->window, step);
  const StepPtrs p = step_ptrs(c);
  // the catalogue pass was done by the forward: only the slabs are combined here (scale, exp(m - lse), + coef * E[a])
  // (the soft part was formed by the forward, right behind the catalogue pass: here only the term that needs the loss)
  CQL_TRY(cql_qhead_dh_finish(w.ws_qb, B, N, d, w.lse, w.coef, w.act, p.Eout_b, alpha_scale(c), w.dH,
                              (hipStream_t)stream, dh_early() ? 2 : 0));
  SideStream& ss = side_stream();
  if (ss.ok) CQL_HIP_TRY(hipEventRecord(ss.dh, (hipStream_t)stream), "train_step_backward_rest");
  return CQLREC_OK;
}

// the chain behind dH: encoder backward, window-gather backward
int backward_chain_impl(const cqlrec_train_ctx* c, uint64_t step, cqlrec_stream stream) {
  const cqlrec_layout& L = c->layout;
  const int32_t B = c->batch, d = L.d, W = c->window;
  const int64_t N = L.n_items;
  StepWs w = carve_step(c->ws, B, N, d, W, step);
  const StepPtrs p = step_ptrs(c);
  // The window-gather backward needs dh0 only: the weight / bias gradients of the encoder (two launches reading dA1) go
  // to the branch stream, idle in the backward, and join before this function returns.  CQL_ENC_SPLIT=0: in line.
  SideStream& ss = side_stream();
  hipStream_t s = (hipStream_t)stream;
  static const int enc_split = !(getenv("CQL_ENC_SPLIT") && getenv("CQL_ENC_SPLIT")[0] == '0');
  const bool split = enc_split && ss.ok && ss.s2 && ss.s2 != s;
  CQL_TRY(cql_encoder_bwd_parts(w.dH, w.zb, w.h0b, p.W1_b, p.W2_b, B, d, w.ws_enc, w.ws_enc_bytes, c->grads + L.off_W1,
                                c->grads + L.off_b1, c->grads + L.off_W2, c->grads + L.off_b2, w.dh0, split ? 1 : 3, s));
  if (split) {
    CQL_HIP_TRY(hipEventRecord(ss.fork2, s), "train_step_backward_rest");
    CQL_HIP_TRY(hipStreamWaitEvent(ss.s2, ss.fork2, 0), "train_step_backward_rest");
    CQL_TRY(cql_encoder_bwd_parts(w.dH, w.zb, w.h0b, p.W1_b, p.W2_b, B, d, w.ws_enc, w.ws_enc_bytes, c->grads + L.off_W1,
                                  c->grads + L.off_b1, c->grads + L.off_W2, c->grads + L.off_b2, w.dh0, 2, ss.s2));
    CQL_HIP_TRY(hipEventRecord(ss.join2, ss.s2), "train_step_backward_rest");
  }
  if (g_mark_phase == 1) mark(MK_ENC_DX, s);
  if (ss.ok) CQL_HIP_TRY(hipStreamWaitEvent(s, ss.sorted[step & 1], 0), "train_step_backward_rest");
  CQL_TRY(cqlrec_gather_pool_bwd_apply(w.dh0, B, W, d, N, w.ws_gb, w.ws_gb_bytes, c->grads + L.off_E_in, stream));
  if (g_mark_phase == 1) mark(MK_GATHER_BWD, s);
  if (split) CQL_HIP_TRY(hipStreamWaitEvent(s, ss.join2, 0), "train_step_backward_rest");
  return CQLREC_OK;
}

int backward_rest_impl(const cqlrec_train_ctx* c, uint64_t step, cqlrec_stream stream) {
  CQL_TRY(backward_states_impl(c, step, stream));
  return backward_chain_impl(c, step, stream);
}
}  // namespace

// phase 1: sample + forward + loss (+ the sort for the gather backward, forked onto the side stream)
extern "C" int cqlrec_train_step_forward(const cqlrec_train_ctx* c, uint64_t step, float* loss_out,
                                         cqlrec_stream stream) {
  CQL_TRY(check_ctx(c));
  return forward_impl(c, step, loss_out, stream, nullptr);
}

extern "C" int cqlrec_train_step_forward_after(const cqlrec_train_ctx* c, uint64_t step, float* loss_out,
                                               cqlrec_stream stream, void* items_ready) {
  CQL_TRY(check_ctx(c));
  return forward_impl(c, step, loss_out, stream, (hipEvent_t)items_ready);
}

// phase 1 + the long kernel of phase 2 (see the header).  The cut pieces of that kernel wait, by step parity, for the
// backward_items call of the same step.
static CqlAdamFix g_early_fix[2];
static bool g_early_long[2] = {false, false};
extern "C" int cqlrec_train_step_forward_early_items(const cqlrec_train_ctx* c, uint64_t step, float* loss_out,
                                                     cqlrec_stream stream, void* items_ready, cqlrec_stream items_stream) {
  CQL_TRY(check_ctx(c));
  SideStream& ss = side_stream();
  const bool early = items_stream && items_stream != stream && need_side_streams(ss) && ss.ok && !onehot_atomic();
  g_early_long[step & 1] = false;
  if (!early) return forward_impl(c, step, loss_out, stream, (hipEvent_t)items_ready);
  g_early_fix[step & 1] = CqlAdamFix{};
  CQL_TRY(forward_impl(c, step, loss_out, stream, (hipEvent_t)items_ready, nullptr, (hipStream_t)items_stream,
                       &g_early_fix[step & 1]));
  g_early_long[step & 1] = true;
  return CQLREC_OK;
}

// phase 2: the catalogue-side gradients g_E_out, g_b_out (half of the gradient bytes; a data-parallel caller starts
// their all-reduce while phase 3 runs)
extern "C" int cqlrec_train_step_backward_items(const cqlrec_train_ctx* c, uint64_t step, cqlrec_stream stream) {
  CQL_TRY(check_ctx(c));
  if (g_early_long[step & 1]) {      // the long kernel is running (or done) on this stream: one-hot rows, then the cut pieces
    g_early_long[step & 1] = false;
    const cqlrec_layout& L = c->layout;
    CQL_TRY(backward_items_onehot_impl(c, step, stream));
    return cql_qde_fixup_deferred(g_early_fix[step & 1], c->grads + L.off_E_out, c->grads + L.off_b_out, (hipStream_t)stream);
  }
  return backward_items_impl(c, step, stream);
}

// phase 3: dH, encoder backward, window-gather backward (joins the side stream)
extern "C" int cqlrec_train_step_backward_rest(const cqlrec_train_ctx* c, uint64_t step, cqlrec_stream stream) {
  CQL_TRY(check_ctx(c));
  return backward_rest_impl(c, step, stream);
}

extern "C" int cqlrec_train_step_fwd_bwd(const cqlrec_train_ctx* c, uint64_t step, float* loss_out,
                                         cqlrec_stream stream) {
  CQL_TRY(cqlrec_train_step_forward(c, step, loss_out, stream));
  // The item-side backward (one long MFMA-bound kernel) runs on its own stream UNDER the state-side chain (dh_finish,
  // encoder and window-gather backward: latency / HBM bound).  Fork after the loss, join before returning.
  SideStream& ss = side_stream();
  hipStream_t s = (hipStream_t)stream;
  if (need_side_streams(ss) && hipEventRecord(ss.loss, s) == hipSuccess && hipStreamWaitEvent(ss.s, ss.loss, 0) == hipSuccess) {
    CQL_TRY(backward_items_impl(c, step, (cqlrec_stream)ss.s));
    CQL_HIP_TRY(hipEventRecord(ss.items, ss.s), "train_step_fwd_bwd");
    CQL_TRY(backward_rest_impl(c, step, stream));
    CQL_HIP_TRY(hipStreamWaitEvent(s, ss.items, 0), "train_step_fwd_bwd");
    return CQLREC_OK;
  }
  CQL_TRY(backward_rest_impl(c, step, stream));
  return backward_items_impl(c, step, stream);
}

// n_steps whole steps (sample .. Adam) of a single-rank job, software-pipelined across the two halves of the model:
//   * side stream: item-side backward (dE_out, db_out) of step t, then Adam on the E_out/b_out range;
//   * main stream: state-side backward (dH, encoder, window gather) of step t, then Adam on E_in + encoder -- HBM-bound,
//     it runs under the MFMA-bound item-side kernel -- then ALREADY the prologue of step t+1 (sample, window gathers,
//     encoder: they read only E_in / W), which waits for the item-side Adam only in front of its Q-head kernels.
// Same dataflow as fwd_bwd + update per step; joined before returning.  world must be 1 (no all-reduce in here).
int update_range_impl(const cqlrec_train_ctx* c, uint64_t step, int64_t lo, int64_t hi, cqlrec_stream stream,
                      const CqlAdamFix* fix);
extern "C" int cqlrec_train_steps(const cqlrec_train_ctx* c, uint64_t step0, int32_t n_steps, float* loss_out,
                                  cqlrec_stream stream) {
  CQL_TRY(check_ctx(c));
  CQL_REQUIRE(n_steps >= 0, "train_steps: n_steps=%d", n_steps);
  CQL_REQUIRE(c->world == 1, "train_steps: world=%d; data-parallel callers use the phase entry points", c->world);
  const cqlrec_layout& L = c->layout;
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t pending = nullptr;   // item-side Adam of the previous step
  hipEvent_t sampled = nullptr;   // transitions + sorted pairs of this step, prepared during the previous backward
  need_side_streams(side_stream(), true);
  for (int32_t i = 0; i < n_steps; ++i) {
    const uint64_t step = step0 + (uint64_t)i;
    // CQL_EARLY_DE=0: the long item-side kernel behind the loss (A/B knob); default: behind the fused forward
    static const int early_de = !(getenv("CQL_EARLY_DE") && getenv("CQL_EARLY_DE")[0] == '0');
    SideStream& ss0 = side_stream();
    const bool early = early_de && ss0.ok && ss0.s && !onehot_atomic();
    CqlAdamFix fix = {};
    // the sum of the loss terms goes to the head of the sample-ahead stream (it has slack) when that stream is used
    bool sum_deferred = false;
    const bool can_defer = ss0.ok && ss0.s3 && i + 1 < n_steps;
    CQL_TRY(forward_impl(c, step, loss_out ? loss_out + i : nullptr, stream, pending, sampled, early ? ss0.s : nullptr, &fix,
                         can_defer ? &sum_deferred : nullptr));
    pending = sampled = nullptr;
    if (g_marks_on && n_steps >= 4) {   // marks: backward of step n/2, forward of step n/2 + 1
      if (i == n_steps / 2) g_mark_phase = 1;
      else if (i == n_steps / 2 + 1) g_mark_phase = 0;
    }
    SideStream& ss = side_stream();
    if (g_mark_phase == 1) mark(MK_LOSS, s);
    if (ss.ok && hipEventRecord(ss.loss, s) == hipSuccess) {
      if (i + 1 < n_steps && hipStreamWaitEvent(ss.s3, ss.loss, 0) == hipSuccess) {
        if (sum_deferred) {
          CQL_TRY(loss_sum_impl(c, step, loss_out ? loss_out + i : nullptr, ss.s3));
          sum_deferred = false;
        }
        // the next step's transitions depend on (seed, step) only: sample them now, on a stream of their own, and sort
        // the pairs its backward will need (two radix sorts, ~35 small launches) behind that
        CQL_TRY(sample_ahead(c, step + 1, (cqlrec_stream)ss.s3, ss.presample));
        CQL_HIP_TRY(hipEventRecord(ss.sorted[(step + 1) & 1], ss.s3), "train_steps");
        if (g_mark_phase == 1) mark(MK_SORT_NEXT, ss.s3);
        sampled = ss.presample;
      }
      if (sum_deferred) CQL_TRY(loss_sum_impl(c, step, loss_out ? loss_out + i : nullptr, s));      // (s3 could not take it)
      CQL_HIP_TRY(hipStreamWaitEvent(ss.s, ss.loss, 0), "train_steps");
      // sparse scatter, then the long dE_out kernel; the sum of its cut pieces is left to the item-side Adam below
      static const int defer_fixup = !(getenv("CQL_DEFER_FIXUP") && getenv("CQL_DEFER_FIXUP")[0] == '0');
      if (early) {          // the long kernel is running (or done) already: the one-hot part on top, cut pieces as asked
        CQL_TRY(backward_items_onehot_impl(c, step, (cqlrec_stream)ss.s));
        if (!defer_fixup) {
          CQL_TRY(cql_qde_fixup_deferred(fix, c->grads + L.off_E_out, c->grads + L.off_b_out, ss.s));
          fix.valid = 0;
        }
      } else {
        CQL_TRY(backward_items_impl(c, step, (cqlrec_stream)ss.s, defer_fixup ? &fix : nullptr));
      }
      if (g_mark_phase == 1) mark(MK_DE, ss.s);
      CQL_TRY(backward_states_impl(c, step, stream));                      // dh_finish, records ss.dh
      if (g_mark_phase == 1) mark(MK_DH, s);
      CQL_TRY(backward_chain_impl(c, step, stream));                       // encoder, window gather
      if (g_mark_phase == 1) mark(MK_CHAIN, s);
      CQL_TRY(cqlrec_train_step_update_range(c, step, L.off_W1, L.total, stream));
      CQL_TRY(cqlrec_train_step_update_range(c, step, 0, L.off_E_out, stream));
      if (g_mark_phase == 1) mark(MK_ADAM_IN, s);
      // The two Adam launches are HBM-bound and ready at about the same time: side by side (default) each takes twice as
      // long and the next prologue (which needs E_in / W only) starts behind both.  CQL_ADAM_SERIAL=1 runs them one after
      // the other, state side first, so that the prologue runs under the item-side Adam -- measured no faster (0.802 vs
      // 0.796 ms per step at cfg3).
      static const int adam_serial = getenv("CQL_ADAM_SERIAL") && getenv("CQL_ADAM_SERIAL")[0] == '1';   // A/B knob
      if (adam_serial) {
        CQL_HIP_TRY(hipEventRecord(ss.adam_in, s), "train_steps");
        CQL_HIP_TRY(hipStreamWaitEvent(ss.s, ss.adam_in, 0), "train_steps");
      }
      CQL_HIP_TRY(hipStreamWaitEvent(ss.s, ss.dh, 0), "train_steps");     // dh_finish reads the E_out shadow
      fix.rows_off = 0;
      fix.cs_off = L.off_b_out - L.off_E_out;
      CQL_TRY(update_range_impl(c, step, L.off_E_out, L.off_W1, (cqlrec_stream)ss.s, &fix));
      CQL_HIP_TRY(hipEventRecord(ss.eout, ss.s), "train_steps");
      if (g_mark_phase == 1) { mark(MK_ADAM_OUT, ss.s); g_mark_phase = 2; }
      pending = ss.eout;
    } else {
      if (sum_deferred) CQL_TRY(loss_sum_impl(c, step, loss_out ? loss_out + i : nullptr, s));
      CQL_TRY(backward_rest_impl(c, step, stream));
      CQL_TRY(backward_items_impl(c, step, stream));
      CQL_TRY(cqlrec_train_step_update_range(c, step, 0, L.total, stream));
    }
  }
  g_mark_phase = 0;
  if (pending) CQL_HIP_TRY(hipStreamWaitEvent(s, pending, 0), "train_steps");
  return CQLREC_OK;
}

// Adam + target + shadows (+ zero grads) over elements [lo, hi) of the flat buffers (multiples of 4)
extern "C" int cqlrec_train_step_update_range(const cqlrec_train_ctx* c, uint64_t step, int64_t lo, int64_t hi,
                                              cqlrec_stream stream) {
  return update_range_impl(c, step, lo, hi, stream, nullptr);
}

int update_range_impl(const cqlrec_train_ctx* c, uint64_t step, int64_t lo, int64_t hi, cqlrec_stream stream,
                      const CqlAdamFix* fix) {
  CQL_TRY(check_ctx(c));
  CQL_REQUIRE(lo >= 0 && hi <= c->layout.total && lo < hi && lo % 4 == 0 && hi % 4 == 0,
              "train_step_update_range: bad range [%lld, %lld)", (long long)lo, (long long)hi);
  const double t = (double)(step + 1);
  const double bc1 = 1.0 - pow((double)c->beta1, t);
  const double bc2 = 1.0 - pow((double)c->beta2, t);
  const float step_size = (float)((double)c->lr / bc1);
  const float sqrt_bc2 = (float)sqrt(bc2);
  return cql_adam_ema_fix(c->theta + lo, c->grads + lo, c->adam_m + lo, c->adam_v + lo, c->target + lo, c->theta_b + lo,
                          c->target_b + lo, hi - lo, step_size, sqrt_bc2, (float)c->beta1, (float)c->beta2, (float)c->eps,
                          (float)c->tau, 1, fix, (hipStream_t)stream);
}

extern "C" int cqlrec_train_step_update(const cqlrec_train_ctx* c, uint64_t step, cqlrec_stream stream) {
  CQL_TRY(check_ctx(c));
  return cqlrec_train_step_update_range(c, step, 0, c->layout.total, stream);
}

extern "C" int cqlrec_train_views_get(const cqlrec_train_ctx* c, uint64_t step, cqlrec_train_views* out) {
  CQL_TRY(check_ctx(c));
  CQL_REQUIRE(out != nullptr, "train_views_get: out is NULL");
  const int32_t B = c->batch, d = c->layout.d;
  StepWs w = carve_step(c->ws, B, c->layout.n_items, d, c->window, step);
  out->users = w.users; out->tpos = w.tpos; out->act = w.act; out->a_star = w.a_star;
  out->rew = w.rew; out->done = w.done; out->q_a = w.q_a; out->lse = w.lse; out->q_targ = w.q_targ;
  out->y = w.y; out->coef = w.coef; out->dH = w.dH; out->dh0 = w.dh0; out->h0_s = w.h0_s;
  out->hb_s = w.hb; out->hb_sn = w.hb + (int64_t)B * d; out->hb_tn = w.hb_t;
  return CQLREC_OK;
}
