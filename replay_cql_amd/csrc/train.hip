// Training-step driver (a8): enqueues the whole CQL step on one stream, no host sync, no allocation.
// Split in two halves so that a data-parallel caller can all-reduce ctx->grads (RCCL over xGMI) in between.
#include <math.h>
#include "common.h"

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
  void *ws_q, *ws_qb, *ws_enc, *ws_gb;
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
  w.ws_q = c.take<char>(w.ws_q_bytes);
  w.ws_qb_bytes = cqlrec_qhead_bwd_ws_bytes(B, N, d);
  w.ws_qb = c.take<char>(w.ws_qb_bytes);
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

extern "C" int64_t cqlrec_train_ws_bytes(int32_t batch, int64_t n_items, int32_t d, int32_t window) {
  return carve_step(nullptr, batch, n_items, d, window).total + 256;
}

#define CQL_TRY(expr)            \
  do {                           \
    int rc__ = (expr);           \
    if (rc__ != CQLREC_OK) return rc__; \
  } while (0)

extern "C" int cqlrec_train_step_fwd_bwd(const cqlrec_train_ctx* c, uint64_t step, float* loss_out,
                                         cqlrec_stream stream) {
  CQL_TRY(check_ctx(c));
  const cqlrec_layout& L = c->layout;
  const int32_t B = c->batch, d = L.d, W = c->window;
  const int64_t N = L.n_items;
  StepWs w = carve_step(c->ws, B, N, d, W);
  const int64_t Bd = (int64_t)B * d;

  const uint16_t* Ein_b = c->theta_b + L.off_E_in;
  const uint16_t* Eout_b = c->theta_b + L.off_E_out;
  const uint16_t* W1_b = c->theta_b + L.off_W1;
  const uint16_t* W2_b = c->theta_b + L.off_W2;
  const float* b_out = c->theta + L.off_b_out;
  const float* b1 = c->theta + L.off_b1;
  const float* b2 = c->theta + L.off_b2;
  const uint16_t* tEin_b = c->target_b + L.off_E_in;
  const uint16_t* tEout_b = c->target_b + L.off_E_out;
  const uint16_t* tW1_b = c->target_b + L.off_W1;
  const uint16_t* tW2_b = c->target_b + L.off_W2;
  const float* tb_out = c->target + L.off_b_out;
  const float* tb1 = c->target + L.off_b1;
  const float* tb2 = c->target + L.off_b2;

  // 1. transitions of this rank's slots of the global step
  CQL_TRY(cqlrec_sample_transitions(c->offsets, c->items, c->rewards, c->n_users, c->seed, step,
                                    (uint64_t)c->rank * (uint64_t)B, B, w.users, w.tpos, w.act, w.rew, w.done, stream));
  // 2. state vectors: s, s' under theta; s' under the target net
  CQL_TRY(cqlrec_gather_pool_fwd(Ein_b, c->offsets, c->items, w.users, w.tpos, 0, B, W, d, w.h0_s, w.h0b, nullptr, stream));
  CQL_TRY(cqlrec_gather_pool_fwd(Ein_b, c->offsets, c->items, w.users, w.tpos, 1, B, W, d, nullptr, w.h0b + Bd, nullptr, stream));
  CQL_TRY(cqlrec_gather_pool_fwd(tEin_b, c->offsets, c->items, w.users, w.tpos, 1, B, W, d, nullptr, w.h0b_t, nullptr, stream));
  // 3. encoder
  CQL_TRY(cqlrec_linear_bf16(w.h0b, W1_b, b1, 2 * (int64_t)B, d, 1, nullptr, w.zb, stream));
  CQL_TRY(cqlrec_linear_bf16(w.zb, W2_b, b2, 2 * (int64_t)B, d, 0, nullptr, w.hb, stream));
  CQL_TRY(cqlrec_linear_bf16(w.h0b_t, tW1_b, tb1, B, d, 1, nullptr, w.zb_t, stream));
  CQL_TRY(cqlrec_linear_bf16(w.zb_t, tW2_b, tb2, B, d, 0, nullptr, w.hb_t, stream));
  // 4. Q-head: logsumexp over the catalog for s, argmax for s'
  CQL_TRY(cqlrec_qhead_fwd(w.hb, B, Eout_b, b_out, N, d, CQLREC_QHEAD_LSE, w.ws_q, w.ws_q_bytes, w.lse, nullptr, w.nlse2, stream));
  CQL_TRY(cqlrec_qhead_fwd(w.hb + Bd, B, Eout_b, b_out, N, d, CQLREC_QHEAD_ARGMAX, w.ws_q, w.ws_q_bytes, w.maxv, w.a_star, nullptr, stream));
  CQL_TRY(cqlrec_gather_dot(w.hb, Eout_b, b_out, w.act, B, d, w.q_a, stream));
  CQL_TRY(cqlrec_gather_dot(w.hb_t, tEout_b, tb_out, w.a_star, B, d, w.q_targ, stream));
  // 5. loss + dQ coefficients
  const float inv_batch = 1.0f / ((float)B * (float)c->world);
  CQL_TRY(cqlrec_td_loss(w.q_a, w.lse, w.q_targ, w.rew, w.done, B, c->gamma, c->alpha, inv_batch, w.coef, w.y,
                         loss_out ? loss_out : w.loss, stream));
  // 6. backward
  CQL_TRY(cqlrec_qhead_bwd(w.hb, w.nlse2, w.coef, w.act, B, Eout_b, b_out, N, d, c->alpha * inv_batch, w.ws_qb,
                           w.ws_qb_bytes, w.dH, c->grads + L.off_E_out, c->grads + L.off_b_out, stream));
  CQL_TRY(cqlrec_encoder_bwd(w.dH, w.zb, w.h0b, W1_b, W2_b, B, d, w.ws_enc, w.ws_enc_bytes, c->grads + L.off_W1,
                             c->grads + L.off_b1, c->grads + L.off_W2, c->grads + L.off_b2, w.dh0, stream));
  CQL_TRY(cqlrec_gather_pool_bwd_sorted(w.dh0, c->offsets, c->items, w.users, w.tpos, 0, B, W, d, N, w.ws_gb,
                                        w.ws_gb_bytes, c->grads + L.off_E_in, stream));
  return CQLREC_OK;
}

extern "C" int cqlrec_train_step_update(const cqlrec_train_ctx* c, uint64_t step, cqlrec_stream stream) {
  CQL_TRY(check_ctx(c));
  const double t = (double)(step + 1);
  const double bc1 = 1.0 - pow((double)c->beta1, t);
  const double bc2 = 1.0 - pow((double)c->beta2, t);
  const float step_size = (float)((double)c->lr / bc1);
  const float sqrt_bc2 = (float)sqrt(bc2);
  return cqlrec_adam_ema(c->theta, c->grads, c->adam_m, c->adam_v, c->target, c->theta_b, c->target_b, c->layout.total,
                         step_size, sqrt_bc2, c->beta1, c->beta2, c->eps, c->tau, 1, stream);
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
