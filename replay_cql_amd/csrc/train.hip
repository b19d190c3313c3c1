// Training-step driver (a8): enqueues the whole CQL step on one stream, no host sync, no allocation.
// Split in two halves so that a data-parallel caller can all-reduce ctx->grads (RCCL over xGMI) in between.
#include <math.h>
#include <stdlib.h>
#include "common.h"

static int g_concurrency = -1;   // -1: read CQL_CONCURRENCY on first use (default on)

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
  float *rew, *done, *q_a, *lse, *nlse2, *q_targ, *y, *coef, *maxv, *loss;
  float *h0_s, *dH, *dh0;
  uint16_t *h0b, *zb, *hb, *h0b_t, *zb_t, *hb_t;
  void *ws_q, *ws_q2, *ws_qb, *ws_qb2, *ws_enc, *ws_gb;
  int64_t ws_q_bytes, ws_qb_bytes, ws_enc_bytes, ws_gb_bytes;
  int64_t total;
};

StepWs carve_step(void* ws, int32_t B, int64_t N, int32_t d, int32_t L) {
  StepWs w;
  Carve c(ws);
  w.users = c.take<int32_t>(B);
  w.tpos = c.take<int32_t>(B);
  w.act = c.take<int32_t>(B);
  w.a_star = c.take<int32_t>(B);
  w.rew = c.take<float>(B);
  w.done = c.take<float>(B);
  w.q_a = c.take<float>(B);
  w.lse = c.take<float>(B);
  w.nlse2 = c.take<float>(B);
  w.q_targ = c.take<float>(B);
  w.y = c.take<float>(B);
  w.coef = c.take<float>(B);
  w.maxv = c.take<float>(B);
  w.loss = c.take<float>(64);
  w.h0_s = c.take<float>((int64_t)B * d);
  w.dH = c.take<float>((int64_t)B * d);
  w.dh0 = c.take<float>((int64_t)B * d);
  w.h0b = c.take<uint16_t>((int64_t)2 * B * d);
  w.zb = c.take<uint16_t>((int64_t)2 * B * d);
  w.hb = c.take<uint16_t>((int64_t)2 * B * d);
  w.h0b_t = c.take<uint16_t>((int64_t)B * d);
  w.zb_t = c.take<uint16_t>((int64_t)B * d);
  w.hb_t = c.take<uint16_t>((int64_t)B * d);
  w.ws_q_bytes = cqlrec_qhead_ws_bytes(B, N, d);
  w.ws_q = c.take<char>(w.ws_q_bytes);     // LSE partials (branch A)
  w.ws_q2 = c.take<char>(w.ws_q_bytes);    // ARGMAX partials (branch B runs concurrently)
  w.ws_qb_bytes = cqlrec_qhead_bwd_ws_bytes(B, N, d);
  w.ws_qb = c.take<char>(w.ws_qb_bytes);    // state-side backward (dH slabs)
  w.ws_qb2 = c.take<char>(w.ws_qb_bytes);   // item-side backward: its own scratch, the two kernels run concurrently
  w.ws_enc_bytes = cqlrec_encoder_bwd_ws_bytes(B, d);
  w.ws_enc = c.take<char>(w.ws_enc_bytes);
  w.ws_gb_bytes = cqlrec_gather_pool_bwd_ws_bytes(B, L, d);
  w.ws_gb = c.take<char>(w.ws_gb_bytes);
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
  hipEvent_t forked = nullptr, joined = nullptr, fork2 = nullptr, join2 = nullptr;
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
SideStream& side_stream() {
  static SideStream ss;
  static SideStream off;   // never ok: serial mode
  if (!concurrency_on()) return off;
  if (!ss.tried) {
    ss.tried = true;
    ss.ok = hipStreamCreateWithFlags(&ss.s, hipStreamNonBlocking) == hipSuccess &&
            hipStreamCreateWithFlags(&ss.s2, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&ss.forked, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ss.joined, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ss.fork2, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ss.join2, hipEventDisableTiming) == hipSuccess;
  }
  return ss;
}

struct StepPtrs {
  const uint16_t *Ein_b, *Eout_b, *W1_b, *W2_b, *tEin_b, *tEout_b, *tW1_b, *tW2_b;
  const float *b_out, *b1, *b2, *tb_out, *tb1, *tb2;
};
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

// phase 1: sample + forward + loss (+ the sort for the gather backward, forked onto the side stream)
extern "C" int cqlrec_train_step_forward(const cqlrec_train_ctx* c, uint64_t step, float* loss_out,
                                         cqlrec_stream stream) {
  CQL_TRY(check_ctx(c));
  const cqlrec_layout& L = c->layout;
  const int32_t B = c->batch, d = L.d, W = c->window;
  const int64_t N = L.n_items;
  StepWs w = carve_step(c->ws, B, N, d, W);
  const StepPtrs p = step_ptrs(c);
  const int64_t Bd = (int64_t)B * d;
  hipStream_t s = (hipStream_t)stream;

  // transitions of this rank's slots of the global step
  CQL_TRY(cqlrec_sample_transitions(c->offsets, c->items, c->rewards, c->n_users, c->seed, step,
                                    (uint64_t)c->rank * (uint64_t)B, B, w.users, w.tpos, w.act, w.rew, w.done, stream));
  SideStream& ss = side_stream();
  if (ss.ok) {
    if (hipEventRecord(ss.forked, s) != hipSuccess || hipStreamWaitEvent(ss.s, ss.forked, 0) != hipSuccess) ss.ok = false;
  }
  if (ss.ok) {
    CQL_TRY(cqlrec_gather_pool_bwd_prepare(c->offsets, c->items, w.users, w.tpos, 0, B, W, d, N, w.ws_gb, w.ws_gb_bytes,
                                           (cqlrec_stream)ss.s));
    if (hipEventRecord(ss.joined, ss.s) != hipSuccess) {
      cql_set_error("train_step_forward: hipEventRecord failed");
      return CQLREC_ERR_HIP;
    }
  } else {
    CQL_TRY(cqlrec_gather_pool_bwd_prepare(c->offsets, c->items, w.users, w.tpos, 0, B, W, d, N, w.ws_gb, w.ws_gb_bytes,
                                           stream));
  }
  // The forward has two independent branches that meet at the TD target:
  //   A (this stream):  s  under theta  -> encoder -> logsumexp over the catalogue, Q(s, a)
  //   B (side stream):  s' under theta  -> encoder -> argmax;  s' under the target net -> encoder -> Q_target(s', a*)
  // Run concurrently, the small gather / encoder / finalize launches of one branch fill the launch gaps of the other.
  cqlrec_stream sb = stream;
  const bool par = ss.ok && hipEventRecord(ss.fork2, s) == hipSuccess && hipStreamWaitEvent(ss.s2, ss.fork2, 0) == hipSuccess;
  if (par) sb = (cqlrec_stream)ss.s2;
  // ---- branch A
  CQL_TRY(cqlrec_gather_pool_fwd(p.Ein_b, c->offsets, c->items, w.users, w.tpos, 0, B, W, d, w.h0_s, w.h0b, nullptr, stream));
  CQL_TRY(cqlrec_linear_bf16(w.h0b, p.W1_b, p.b1, B, d, 1, nullptr, w.zb, stream));
  CQL_TRY(cqlrec_linear_bf16(w.zb, p.W2_b, p.b2, B, d, 0, nullptr, w.hb, stream));
  CQL_TRY(cqlrec_qhead_fwd(w.hb, B, p.Eout_b, p.b_out, N, d, CQLREC_QHEAD_LSE, w.ws_q, w.ws_q_bytes, w.lse, nullptr, w.nlse2, stream));
  CQL_TRY(cqlrec_gather_dot(w.hb, p.Eout_b, p.b_out, w.act, B, d, w.q_a, stream));
  // ---- branch B
  CQL_TRY(cqlrec_gather_pool_fwd(p.Ein_b, c->offsets, c->items, w.users, w.tpos, 1, B, W, d, nullptr, w.h0b + Bd, nullptr, sb));
  CQL_TRY(cqlrec_gather_pool_fwd(p.tEin_b, c->offsets, c->items, w.users, w.tpos, 1, B, W, d, nullptr, w.h0b_t, nullptr, sb));
  CQL_TRY(cqlrec_linear_bf16(w.h0b + Bd, p.W1_b, p.b1, B, d, 1, nullptr, w.zb + Bd, sb));
  CQL_TRY(cqlrec_linear_bf16(w.zb + Bd, p.W2_b, p.b2, B, d, 0, nullptr, w.hb + Bd, sb));
  CQL_TRY(cqlrec_linear_bf16(w.h0b_t, p.tW1_b, p.tb1, B, d, 1, nullptr, w.zb_t, sb));
  CQL_TRY(cqlrec_linear_bf16(w.zb_t, p.tW2_b, p.tb2, B, d, 0, nullptr, w.hb_t, sb));
  CQL_TRY(cqlrec_qhead_fwd(w.hb + Bd, B, p.Eout_b, p.b_out, N, d, CQLREC_QHEAD_ARGMAX, w.ws_q2, w.ws_q_bytes, w.maxv, w.a_star, nullptr, sb));
  CQL_TRY(cqlrec_gather_dot(w.hb_t, p.tEout_b, p.tb_out, w.a_star, B, d, w.q_targ, sb));
  if (par && (hipEventRecord(ss.join2, ss.s2) != hipSuccess || hipStreamWaitEvent(s, ss.join2, 0) != hipSuccess)) {
    cql_set_error("train_step_forward: joining the forward branches failed");
    return CQLREC_ERR_HIP;
  }
  // loss + dQ coefficients
  const float inv_batch = 1.0f / ((float)B * (float)c->world);
  CQL_TRY(cqlrec_td_loss(w.q_a, w.lse, w.q_targ, w.rew, w.done, B, c->gamma, c->alpha, inv_batch, w.coef, w.y,
                         loss_out ? loss_out : w.loss, stream));
  return CQLREC_OK;
}

// phase 2: the catalogue-side gradients g_E_out, g_b_out (half of the gradient bytes; a data-parallel caller starts
// their all-reduce while phase 3 runs)
extern "C" int cqlrec_train_step_backward_items(const cqlrec_train_ctx* c, uint64_t step, cqlrec_stream stream) {
  (void)step;
  CQL_TRY(check_ctx(c));
  const cqlrec_layout& L = c->layout;
  const int32_t B = c->batch, d = L.d;
  StepWs w = carve_step(c->ws, B, L.n_items, d, c->window);
  const StepPtrs p = step_ptrs(c);
  const float inv_batch = 1.0f / ((float)B * (float)c->world);
  return cqlrec_qhead_bwd_items(w.hb, w.nlse2, w.coef, w.act, B, p.Eout_b, p.b_out, L.n_items, d, c->alpha * inv_batch,
                                w.ws_qb2, w.ws_qb_bytes, c->grads + L.off_E_out, c->grads + L.off_b_out, stream);
}

// phase 3: dH, encoder backward, window-gather backward (joins the side stream)
extern "C" int cqlrec_train_step_backward_rest(const cqlrec_train_ctx* c, uint64_t step, cqlrec_stream stream) {
  (void)step;
  CQL_TRY(check_ctx(c));
  const cqlrec_layout& L = c->layout;
  const int32_t B = c->batch, d = L.d, W = c->window;
  const int64_t N = L.n_items;
  StepWs w = carve_step(c->ws, B, N, d, W);
  const StepPtrs p = step_ptrs(c);
  const float inv_batch = 1.0f / ((float)B * (float)c->world);
  CQL_TRY(cqlrec_qhead_bwd_states(w.hb, w.nlse2, w.coef, w.act, B, p.Eout_b, p.b_out, N, d, c->alpha * inv_batch, w.ws_qb,
                                  w.ws_qb_bytes, w.dH, stream));
  CQL_TRY(cqlrec_encoder_bwd(w.dH, w.zb, w.h0b, p.W1_b, p.W2_b, B, d, w.ws_enc, w.ws_enc_bytes, c->grads + L.off_W1,
                             c->grads + L.off_b1, c->grads + L.off_W2, c->grads + L.off_b2, w.dh0, stream));
  SideStream& ss = side_stream();
  if (ss.ok && hipStreamWaitEvent((hipStream_t)stream, ss.joined, 0) != hipSuccess) {
    cql_set_error("train_step_backward_rest: hipStreamWaitEvent failed");
    return CQLREC_ERR_HIP;
  }
  CQL_TRY(cqlrec_gather_pool_bwd_apply(w.dh0, B, W, d, N, w.ws_gb, w.ws_gb_bytes, c->grads + L.off_E_in, stream));
  return CQLREC_OK;
}

extern "C" int cqlrec_train_step_fwd_bwd(const cqlrec_train_ctx* c, uint64_t step, float* loss_out,
                                         cqlrec_stream stream) {
  CQL_TRY(cqlrec_train_step_forward(c, step, loss_out, stream));
  // The item-side and the state-side halves of the Q-head backward are independent.  Each alone leaves part of the
  // chip idle in its last "round" of resident blocks (782 resp. 512 blocks on 512 slots at cfg3); issued on two
  // streams, the hardware scheduler packs them.  Fork after the loss, join before returning.
  SideStream& ss = side_stream();
  hipStream_t s = (hipStream_t)stream;
  static hipEvent_t ev_loss = nullptr, ev_items = nullptr;
  if (ss.ok && !ev_loss) {
    if (hipEventCreateWithFlags(&ev_loss, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ev_items, hipEventDisableTiming) != hipSuccess)
      ss.ok = false;
  }
  if (ss.ok && hipEventRecord(ev_loss, s) == hipSuccess && hipStreamWaitEvent(ss.s, ev_loss, 0) == hipSuccess) {
    CQL_TRY(cqlrec_train_step_backward_items(c, step, (cqlrec_stream)ss.s));
    if (hipEventRecord(ev_items, ss.s) != hipSuccess) {
      cql_set_error("train_step_fwd_bwd: hipEventRecord failed");
      return CQLREC_ERR_HIP;
    }
    CQL_TRY(cqlrec_train_step_backward_rest(c, step, stream));
    if (hipStreamWaitEvent(s, ev_items, 0) != hipSuccess) {
      cql_set_error("train_step_fwd_bwd: hipStreamWaitEvent failed");
      return CQLREC_ERR_HIP;
    }
    return CQLREC_OK;
  }
  CQL_TRY(cqlrec_train_step_backward_items(c, step, stream));
  return cqlrec_train_step_backward_rest(c, step, stream);
}

// Adam + target + shadows (+ zero grads) over elements [lo, hi) of the flat buffers (multiples of 4)
extern "C" int cqlrec_train_step_update_range(const cqlrec_train_ctx* c, uint64_t step, int64_t lo, int64_t hi,
                                              cqlrec_stream stream) {
  CQL_TRY(check_ctx(c));
  CQL_REQUIRE(lo >= 0 && hi <= c->layout.total && lo < hi && lo % 4 == 0 && hi % 4 == 0,
              "train_step_update_range: bad range [%lld, %lld)", (long long)lo, (long long)hi);
  const double t = (double)(step + 1);
  const double bc1 = 1.0 - pow((double)c->beta1, t);
  const double bc2 = 1.0 - pow((double)c->beta2, t);
  const float step_size = (float)((double)c->lr / bc1);
  const float sqrt_bc2 = (float)sqrt(bc2);
  return cqlrec_adam_ema(c->theta + lo, c->grads + lo, c->adam_m + lo, c->adam_v + lo, c->target + lo, c->theta_b + lo,
                         c->target_b + lo, hi - lo, step_size, sqrt_bc2, c->beta1, c->beta2, c->eps, c->tau, 1, stream);
}

extern "C" int cqlrec_train_step_update(const cqlrec_train_ctx* c, uint64_t step, cqlrec_stream stream) {
  CQL_TRY(check_ctx(c));
  return cqlrec_train_step_update_range(c, step, 0, c->layout.total, stream);
}

extern "C" int cqlrec_train_views_get(const cqlrec_train_ctx* c, cqlrec_train_views* out) {
  CQL_TRY(check_ctx(c));
  CQL_REQUIRE(out != nullptr, "train_views_get: out is NULL");
  const int32_t B = c->batch, d = c->layout.d;
  StepWs w = carve_step(c->ws, B, c->layout.n_items, d, c->window);
  out->users = w.users; out->tpos = w.tpos; out->act = w.act; out->a_star = w.a_star;
  out->rew = w.rew; out->done = w.done; out->q_a = w.q_a; out->lse = w.lse; out->q_targ = w.q_targ;
  out->y = w.y; out->coef = w.coef; out->dH = w.dH; out->dh0 = w.dh0; out->h0_s = w.h0_s;
  out->hb_s = w.hb; out->hb_sn = w.hb + (int64_t)B * d; out->hb_tn = w.hb_t;
  return CQLREC_OK;
}
