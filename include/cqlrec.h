/*
 * cqlrec.h -- C ABI of libcqlrec.so, the MI355X (gfx950) CQL recommender hot path.
 *
 * The reference (monkey0head/RePlay_cql @ 2025-02-28 = replay-rec 0.10.0) has NO FFI layer and no CQL model
 * (SURVEY.md F1-F3): its plug-in boundary is the Python abstract class replay/models/base_rec.py:1202-1335
 * (Recommender.fit/predict) with the hooks _fit (:376-392) and _predict (:607-637).  The entry points below
 * are what a `replay/models/cql.py` behind that boundary binds with ctypes; each one cites
 * the reference code whose role it takes over.  Arithmetic follows SURVEY.md section 8.0 (S1-S7, P1-P4).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'd / a torch tensor's data_ptr()) unless marked [host];
 *   - plain pointers and sizes only, no torch types; `stream` is a hipStream_t passed as void* (0 = null stream);
 *   - all calls are asynchronous on `stream`, allocate nothing, never synchronise: they may be captured in a hipGraph
 *     (after one warm-up call: kernels with > 64 KiB of dynamic LDS opt in on first use; tests/test_gpu_kernels.py
 *     ::test_entry_points_capture_in_a_hip_graph).  The step drivers take the step number by value -- sampling seed and
 *     Adam bias corrections are baked into their launches -- so a captured training step replays THAT step;
 *   - bf16 tensors are uint16_t bit patterns; matrices are row-major; W1/W2 are stored [out][in];
 *   - return value: CQLREC_OK, or a negative code with the message available from cqlrec_last_error();
 *   - the kernel-level entry points are re-entrant; the step driver (cqlrec_train_step*) keeps one set of internal
 *     side streams and events per DEVICE (indexed by the calling thread's current device): drive one training context
 *     per device from one host thread at a time (one process per GPU is the deployment model);
 *   - the training step contains no float atomics: the same calls on the same inputs give the same bits.
 */
#ifndef CQLREC_H
#define CQLREC_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CQLREC_OK 0
#define CQLREC_ERR_INVALID (-1)
#define CQLREC_ERR_HIP (-2)
#define CQLREC_ABI_VERSION 2
#define CQLREC_SEG_ALIGN 64 /* every parameter segment starts on a multiple of 64 elements */

typedef void* cqlrec_stream;

int cqlrec_abi_version(void);
const char* cqlrec_last_error(void); /* [host] thread-local message of the last failing call */

/* ---------------------------------------------------------------------------------------------------------
 * Parameter layout (S3).  One flat buffer holds  E_in[(N+1) x d] | E_out[N x d] | b_out[N] | W1[d x d] | b1[d] |
 * W2[d x d] | b2[d]; theta (fp32), grads, Adam m/v, the target copy and both bf16 shadows share these offsets.
 * Replaces the nn.Module parameter containers of the torch analogues (replay/models/neuromf.py:37-211).
 * --------------------------------------------------------------------------------------------------------- */
typedef struct cqlrec_layout {
  int64_t n_items;
  int32_t d;
  int32_t reserved;
  int64_t off_E_in, off_E_out, off_b_out, off_W1, off_b1, off_W2, off_b2;
  int64_t total; /* elements, padded */
} cqlrec_layout;
int cqlrec_layout_make(int64_t n_items, int32_t d, cqlrec_layout* out /* [host] */);

/* ---------------------------------------------------------------------------------------------------------
 * a2/a8  Transition sampler: counter-based, bit-exact with oracle.sample_positions().  Takes the place of the
 * DataLoader iteration in TorchRecommender.train (replay/models/base_torch_rec.py:78-80).
 *   k1 = mix64(seed ^ step*0xD1B54A32D192ED03); p = mulhi64(mix64(k1 + slot0 + b), nnz); user = last u with
 *   offsets[u] <= p; t = p - offsets[u]; act = items[p]; rew = rewards[p]; done = (t == count(u)-1).
 * --------------------------------------------------------------------------------------------------------- */
int cqlrec_sample_transitions(const int64_t* offsets, const int32_t* items, const float* rewards, int64_t n_users,
                              uint64_t seed, uint64_t step, uint64_t slot0, int32_t batch, int32_t* users,
                              int32_t* tpos, int32_t* act, float* rew, float* done, cqlrec_stream stream);

/* ---------------------------------------------------------------------------------------------------------
 * a3  Window gather + masked mean (S4 h0).  No reference counterpart (closest: nn.Embedding lookups,
 * replay/models/neuromf.py:72-73).  State i = the last min(end_i, L) items of users[i] before position
 * end_i = ends[i] + end_delta  (ends == NULL: end_i = row length, the predict-time state, S7).
 * Outputs (either may be NULL): h0 fp32 [n x d], h0_b bf16 [n x d], lens int32 [n].
 * --------------------------------------------------------------------------------------------------------- */
int cqlrec_gather_pool_fwd(const uint16_t* E_in_b, const int64_t* offsets, const int32_t* items,
                           const int32_t* users, const int32_t* ends, int32_t end_delta, int64_t n_states,
                           int32_t L, int32_t d, float* h0, uint16_t* h0_b, int32_t* lens, cqlrec_stream stream);
/* backward of the above: g_E_in[item] += dh0[i] / len_i for every item of window i.
 * cqlrec_gather_pool_bwd: direct fp32 atomic scatter (small inputs / reference form).
 * cqlrec_gather_pool_bwd_sorted: the production form -- (item, state) pairs are radix-sorted by item and summed
 * per run in registers, so a hot item costs one row-add per 64 contributions; rows touched by a single chunk are
 * written with plain stores.  g_E_in must be zero on entry for both.  ws: cqlrec_gather_pool_bwd_ws_bytes. */
int cqlrec_gather_pool_bwd(const float* dh0, const int64_t* offsets, const int32_t* items, const int32_t* users,
                           const int32_t* ends, int32_t end_delta, int64_t n_states, int32_t L, int32_t d,
                           float* g_E_in, cqlrec_stream stream);
int64_t cqlrec_gather_pool_bwd_ws_bytes(int64_t n_states, int32_t L, int32_t d);
/* the same, split at the point where the gradient is needed: _prepare (pairs + sort) depends only on the sampled
 * states, _apply (scale + segmented sum) on dh0.  The step driver runs _prepare on a side stream. */
int cqlrec_gather_pool_bwd_prepare(const int64_t* offsets, const int32_t* items, const int32_t* users,
                                   const int32_t* ends, int32_t end_delta, int64_t n_states, int32_t L, int32_t d,
                                   int64_t n_items, void* ws, int64_t ws_bytes, cqlrec_stream stream);
int cqlrec_gather_pool_bwd_apply(const float* dh0, int64_t n_states, int32_t L, int32_t d, int64_t n_items, void* ws,
                                 int64_t ws_bytes, float* g_E_in, cqlrec_stream stream);
int cqlrec_gather_pool_bwd_sorted(const float* dh0, const int64_t* offsets, const int32_t* items,
                                  const int32_t* users, const int32_t* ends, int32_t end_delta, int64_t n_states,
                                  int32_t L, int32_t d, int64_t n_items, void* ws, int64_t ws_bytes,
                                  float* g_E_in, cqlrec_stream stream);

/* ---------------------------------------------------------------------------------------------------------
 * a4  One encoder layer  Y = act(X_b W_b^T + bias)  on bf16 MFMA, fp32 accumulate (closest reference code:
 * MLP.forward, replay/models/neuromf.py:133-145).  Y (fp32) and Y_b (bf16) may each be NULL.
 * --------------------------------------------------------------------------------------------------------- */
int cqlrec_linear_bf16(const uint16_t* X_b, const uint16_t* W_b, const float* bias, int64_t rows, int32_t d,
                       int32_t relu, float* Y, uint16_t* Y_b, cqlrec_stream stream);
/* Both layers of the state encoder in one launch:  Z_b = bf16(relu(X_b W1_b^T + b1)),  H_b = bf16(Z_b W2_b^T + b2) -- the
 * bits of two cqlrec_linear_bf16 calls (same products, same k-order, same roundings); Z_b is kept for the backward. */
int cqlrec_encoder_fwd(const uint16_t* X_b, const uint16_t* W1_b, const float* b1, const uint16_t* W2_b, const float* b2,
                       int64_t rows, int32_t d, uint16_t* Z_b, uint16_t* H_b, cqlrec_stream stream);
/* fp32 backward of the two-layer encoder on the bf16-valued forward operands.  ws: cqlrec_encoder_bwd_ws_bytes. */
int64_t cqlrec_encoder_bwd_ws_bytes(int64_t rows, int32_t d);
int cqlrec_encoder_bwd(const float* dH, const uint16_t* z_b, const uint16_t* h0_b, const uint16_t* W1_b,
                       const uint16_t* W2_b, int64_t rows, int32_t d, void* ws, int64_t ws_bytes, float* g_W1,
                       float* g_b1, float* g_W2, float* g_b2, float* dh0, cqlrec_stream stream);

/* ---------------------------------------------------------------------------------------------------------
 * a5  Full-catalog Q-head, fused with its row reduction; the rows x N score matrix never reaches HBM.
 * Closest reference code: the catalog-wide Linear + log_softmax of MultVAE (replay/models/mult_vae.py:101,
 * :275-276).   Q[r][j] = <H_b[r], E_out_b[j]> + b_out[j].
 *   mode CQLREC_QHEAD_LSE    : out_val[r] = logsumexp_j Q[r][j]; out_nlse2[r] = -out_val[r]*log2(e) (may be NULL)
 *   mode CQLREC_QHEAD_ARGMAX : out_val[r] = max_j Q[r][j]; out_idx[r] = argmax (ties -> smallest j)
 * ws: cqlrec_qhead_ws_bytes(rows, n_items, d) bytes of scratch.
 * --------------------------------------------------------------------------------------------------------- */
#define CQLREC_QHEAD_LSE 1
#define CQLREC_QHEAD_ARGMAX 2
int64_t cqlrec_qhead_ws_bytes(int64_t rows, int64_t n_items, int32_t d);
int cqlrec_qhead_fwd(const uint16_t* H_b, int64_t rows, const uint16_t* E_out_b, const float* b_out,
                     int64_t n_items, int32_t d, int32_t mode, void* ws, int64_t ws_bytes, float* out_val,
                     int32_t* out_idx, float* out_nlse2, cqlrec_stream stream);

/* out[r] = <H_b[r], E_b[idx[r]]> + b[idx[r]]   (q_a, the target-net Q(s',a*), and _predict_pairs, a11:
 * replay/models/base_rec.py:784-823) */
int cqlrec_gather_dot(const uint16_t* H_b, const uint16_t* E_b, const float* b, const int32_t* idx, int64_t rows,
                      int32_t d, float* out, cqlrec_stream stream);

/* ---------------------------------------------------------------------------------------------------------
 * a6  CQL + double-Q TD loss (S5) and its backward.  Reference analogue of the role: the _loss hooks +
 * loss.backward() (replay/models/base_torch_rec.py:35-37).
 *   y = rew + gamma (1-done) q_targ;  delta = q_a - y;  loss = inv_batch * sum_b [0.5 delta^2 + alpha (lse - q_a)]
 *   coef[b] = (delta - alpha) * inv_batch         (inv_batch = 1 / global batch)
 * loss_out: one float, written (deterministic tree reduction).
 * --------------------------------------------------------------------------------------------------------- */
int cqlrec_td_loss(const float* q_a, const float* lse, const float* q_targ, const float* rew, const float* done,
                   int32_t batch, float gamma, float alpha, float inv_batch, float* coef, float* y,
                   float* loss_out, cqlrec_stream stream);
/* dQ = scale * bf16(exp(Q - lse)) + coef * onehot(act)   (scale = alpha * inv_batch), never materialised:
 *   dH[b]        = scale * sum_j P_b[b][j] E_out_b[j] + coef[b] E_out_b[act[b]]
 *   g_E_out[j]   = scale * sum_b P_b[b][j] H_b[b]     + sum_{b: act[b]=j} coef[b] H_b[b]        (overwritten)
 *   g_b_out[j]   = scale * sum_b P[b][j]              + sum_{b: act[b]=j} coef[b]               (overwritten)
 * nlse2 = -lse*log2(e) from cqlrec_qhead_fwd.  ws: cqlrec_qhead_bwd_ws_bytes. */
int64_t cqlrec_qhead_bwd_ws_bytes(int64_t batch, int64_t n_items, int32_t d);
int cqlrec_qhead_bwd(const uint16_t* H_b, const float* nlse2, const float* coef, const int32_t* act,
                     int64_t batch, const uint16_t* E_out_b, const float* b_out, int64_t n_items, int32_t d,
                     float scale, void* ws, int64_t ws_bytes, float* dH, float* g_E_out, float* g_b_out,
                     cqlrec_stream stream);

/* the two halves of cqlrec_qhead_bwd, separately callable (a data-parallel caller all-reduces g_E_out/g_b_out
 * while the state-side half still runs) */
int cqlrec_qhead_bwd_items(const uint16_t* H_b, const float* nlse2, const float* coef, const int32_t* act,
                           int64_t batch, const uint16_t* E_out_b, const float* b_out, int64_t n_items, int32_t d,
                           float scale, void* ws, int64_t ws_bytes, float* g_E_out, float* g_b_out,
                           cqlrec_stream stream);
int cqlrec_qhead_bwd_states(const uint16_t* H_b, const float* nlse2, const float* coef, const int32_t* act,
                            int64_t batch, const uint16_t* E_out_b, const float* b_out, int64_t n_items, int32_t d,
                            float scale, void* ws, int64_t ws_bytes, float* dH, cqlrec_stream stream);

/* ---------------------------------------------------------------------------------------------------------
 * a5+a10 fused ("flash" form), used by the training step: ONE pass over the catalogue yields per state the
 * logsumexp AND sum_j exp(S_j - m) E_out_b[j] relative to a running reference m (kept in `ws`, one slab per
 * catalogue slice), so the state-side backward needs no second catalogue pass.  Once the TD coefficients are
 * known, dh_finish combines the slabs:
 *     dH[b] = scale * sum_k slab_k[b] * exp(m_k[b] - lse[b]) + coef[b] * E_out_b[act[b]]
 * P is rounded to bf16 relative to the running reference instead of the final lse (same 2^-9 relative error per
 * probability, different rounding points): dH agrees with cqlrec_qhead_bwd_states to ~1e-3 normwise, lse to 1e-6.
 * Replaces, together with cqlrec_qhead_bwd_items, loss.backward() through the catalogue-wide Linear + log_softmax
 * (replay/models/mult_vae.py:101, :274-284; replay/models/base_torch_rec.py:37).
 * --------------------------------------------------------------------------------------------------------- */
int64_t cqlrec_qhead_fused_ws_bytes(int64_t rows, int64_t n_items, int32_t d);
int cqlrec_qhead_fwd_lse_dh(const uint16_t* H_b, int64_t rows, const uint16_t* E_out_b, const float* b_out,
                            int64_t n_items, int32_t d, void* ws, int64_t ws_bytes, float* out_lse,
                            float* out_nlse2 /* -lse*log2(e), may be NULL */, cqlrec_stream stream);
int cqlrec_qhead_dh_finish(const void* ws, int64_t rows, int64_t n_items, int32_t d, const float* lse,
                           const float* coef, const int32_t* act, const uint16_t* E_out_b, float scale, float* dH,
                           cqlrec_stream stream);

/* ---------------------------------------------------------------------------------------------------------
 * a7  Fused Adam + Polyak target + bf16 shadows over the flat buffer (S6).  Replaces optimizer.step()
 * (replay/models/base_torch_rec.py:38; torch.optim.Adam at replay/models/neuromf.py:351-355).
 *   m = b1 m + (1-b1) g;  v = b2 v + ((1-b2) g) g;  theta -= step_size * (m / (sqrt(v)/sqrt_bc2 + eps));
 *   target = (1-tau) target + tau theta;  theta_b = bf16(theta);  target_b = bf16(target);  g = 0 if zero_grads.
 * step_size = lr/(1-b1^t), sqrt_bc2 = sqrt(1-b2^t) are computed by the caller in double and passed as float.
 * --------------------------------------------------------------------------------------------------------- */
int cqlrec_adam_ema(float* theta, float* grads, float* m, float* v, float* target, uint16_t* theta_b,
                    uint16_t* target_b, int64_t n, float step_size, float sqrt_bc2, float beta1, float beta2,
                    float eps, float tau, int32_t zero_grads, cqlrec_stream stream);
/* dst_b = bf16(src) (shadow refresh after load / init) */
int cqlrec_cast_bf16(const float* src, uint16_t* dst_b, int64_t n, cqlrec_stream stream);

/* ---------------------------------------------------------------------------------------------------------
 * a10  All-users top-K scoring (S7).  Takes the place of TorchRecommender._predict's per-user pandas UDF
 * (replay/models/base_torch_rec.py:120-149) + NeuroMF._predict_by_user's argsort (replay/models/neuromf.py:
 * 393-438) + _filter_seen / get_top_k_recs (replay/models/base_rec.py:417-464, replay/utils.py:112-127).
 *   H_b      [n_users x d]  bf16 state vectors (encoder output)
 *   E_b, b   candidate item table [n_cand x d] bf16 and bias [n_cand]
 *   item_ids [n_cand] global id of candidate row c (NULL: identity); must be ascending
 *   seen_off, seen_items: CSR of ascending global item ids to exclude; scored user u uses CSR row
 *            seen_rows[u] (seen_rows NULL: row u).  seen_off NULL: no filter.
 * Output, per user: the k best (score desc, item id asc) admissible items: out_idx/out_val [n_users x k]
 * (padding: -1 / -inf) and out_cnt [n_users] = number of valid entries.
 * --------------------------------------------------------------------------------------------------------- */
int64_t cqlrec_topk_ws_bytes(int64_t n_users, int64_t n_cand, int32_t d, int32_t k);
int cqlrec_score_topk(const uint16_t* H_b, int64_t n_users, const uint16_t* E_b, const float* b, int64_t n_cand,
                      int32_t d, const int32_t* item_ids, const int64_t* seen_off, const int32_t* seen_items,
                      const int32_t* seen_rows, int32_t k, void* ws, int64_t ws_bytes, int32_t* out_idx,
                      float* out_val, int32_t* out_cnt, cqlrec_stream stream);
/* The same in two phases, so that the part that depends on the seen lists only (their bitmap, built in `ws`: it is
 * what _filter_seen's anti-join becomes here) can run on another stream while the caller still encodes the state
 * vectors:  CQLREC_TOPK_SEEN (H_b, outputs ignored; may be NULL) ... CQLREC_TOPK_SCORE on the same ws, ordered
 * behind it by the caller.  CQLREC_TOPK_ALL = cqlrec_score_topk.  Shapes whose selection kernel filters on the
 * fly do nothing in the first phase. */
#define CQLREC_TOPK_ALL 0
#define CQLREC_TOPK_SEEN 1
#define CQLREC_TOPK_SCORE 2
#define CQLREC_TOPK_SEEN_BESIDE 3 /* = _SEEN, launched while a _SCORE phase of another workspace runs on another stream:
                                     the builder then uses what that kernel leaves free of a CU (small LDS tiles) */
int cqlrec_score_topk_phase(const uint16_t* H_b, int64_t n_users, const uint16_t* E_b, const float* b, int64_t n_cand,
                            int32_t d, const int32_t* item_ids, const int64_t* seen_off, const int32_t* seen_items,
                            const int32_t* seen_rows, int32_t k, void* ws, int64_t ws_bytes, int32_t* out_idx,
                            float* out_val, int32_t* out_cnt, int32_t phase, cqlrec_stream stream);

/* Debug / tests: which form of the seen filter the last SEEN (or ALL) phase on `ws` left for the on-chip-selection
 * kernel.  *out (host) = -1: this shape filters by bitmap only; 0: entry lists (256-byte slots per 128 users x 64 items
 * + an overflow area, built in the bitmap's space; the bitmap was not built); 1: the lists did not fit, bitmap.
 * Synchronises `stream`. */
int cqlrec_topk_seen_form(const void* ws, int64_t n_users, int64_t n_cand, int32_t d, int32_t k, int32_t* out,
                          cqlrec_stream stream);

/* ---------------------------------------------------------------------------------------------------------
 * a8  Whole training step = TorchRecommender._run_train_step (replay/models/base_torch_rec.py:32-39) without
 * the per-step host sync.  The step is split in two so that a data-parallel caller can all-reduce ctx.grads
 * (RCCL) between them; losses[] receives one float per step.
 * --------------------------------------------------------------------------------------------------------- */
typedef struct cqlrec_train_ctx {
  cqlrec_layout layout;
  /* training set (CSR by user, S2) */
  const int64_t* offsets;
  const int32_t* items;
  const float* rewards;
  int64_t n_users;
  /* model state, each `layout.total` elements */
  float* theta;
  float* grads;
  float* adam_m;
  float* adam_v;
  float* target;
  uint16_t* theta_b;
  uint16_t* target_b;
  /* hyper-parameters */
  int32_t batch;        /* transitions per step on this rank (multiple of 32) */
  int32_t window;       /* L */
  int32_t world;        /* data-parallel ranks (loss is a mean over batch*world) */
  int32_t rank;
  /* doubles: the Adam bias corrections 1 - beta^t are formed in double from the caller's values (as torch.optim.Adam
   * does with its Python floats); the kernels receive them rounded to float */
  double gamma, alpha, lr, beta1, beta2, eps, tau;
  uint64_t seed;
  /* scratch: cqlrec_train_ws_bytes(batch, n_items, d, window) bytes */
  void* ws;
  int64_t ws_bytes;
} cqlrec_train_ctx;
int64_t cqlrec_train_ws_bytes(int32_t batch, int64_t n_items, int32_t d, int32_t window);
/* sample + forward + loss + backward into ctx->grads (which must be zero on entry; step_update re-zeroes it).
 * loss_out: device float (may be NULL). */
int cqlrec_train_step_fwd_bwd(const cqlrec_train_ctx* ctx /* [host] */, uint64_t step, float* loss_out,
                              cqlrec_stream stream);
/* Adam + target + shadows (+ zero grads). `step` is the same 0-based counter. */
int cqlrec_train_step_update(const cqlrec_train_ctx* ctx /* [host] */, uint64_t step, cqlrec_stream stream);
/* Finer-grained phases for overlapping the gradient all-reduce with compute (fwd_bwd == forward; backward_items;
 * backward_rest.  update == update_range over [0, layout.total)):
 *   forward         sample, state vectors, encoder, Q-head LSE/argmax, TD target, loss
 *   backward_items  g_E_out, g_b_out                (elements [off_E_out, off_W1) of ctx->grads are final afterwards)
 *   backward_rest   dH, encoder gradients, g_E_in   (the remaining elements are final afterwards)
 *   update_range    Adam/Polyak/shadows/zero-grads over elements [lo, hi) (multiples of 4) */
int cqlrec_train_step_forward(const cqlrec_train_ctx* ctx /* [host] */, uint64_t step, float* loss_out,
                              cqlrec_stream stream);
/* The same with a dependency the caller still has in flight on ANOTHER stream: `items_ready` (a hipEvent_t, or NULL)
 * completes when the item-side parameters (E_out, b_out, their shadows and target copies) are up to date.  Sampling,
 * window gathers and both encoders -- they read only E_in / W1 / W2 -- are enqueued in front of the wait, only the
 * catalogue-wide kernels behind it.  A data-parallel caller uses it to start step t+1 while the all-reduce and Adam of
 * the item-side half of step t are still running on its side stream. */
int cqlrec_train_step_forward_after(const cqlrec_train_ctx* ctx /* [host] */, uint64_t step, float* loss_out,
                                    cqlrec_stream stream, void* items_ready /* hipEvent_t */);
/* forward_after that ALSO starts the long part of backward_items -- the dE_out kernel, which needs nothing from the
 * loss but the forward's logsumexp -- on `items_stream` as soon as the catalogue pass of branch A is done, i.e. under
 * the arg-max pass, the TD target and the loss (what cqlrec_train_steps does for a single rank).  The
 * cqlrec_train_step_backward_items call of the same step, which must then be made on `items_stream`, adds only the
 * parts that need the loss (one-hot rows, then the cut pieces: the order in which a gradient row is summed does not
 * change).  items_ready may be NULL. */
int cqlrec_train_step_forward_early_items(const cqlrec_train_ctx* ctx /* [host] */, uint64_t step, float* loss_out,
                                          cqlrec_stream stream, void* items_ready /* hipEvent_t */,
                                          cqlrec_stream items_stream);
int cqlrec_train_step_backward_items(const cqlrec_train_ctx* ctx /* [host] */, uint64_t step, cqlrec_stream stream);
int cqlrec_train_step_backward_rest(const cqlrec_train_ctx* ctx /* [host] */, uint64_t step, cqlrec_stream stream);
int cqlrec_train_step_update_range(const cqlrec_train_ctx* ctx /* [host] */, uint64_t step, int64_t lo, int64_t hi,
                                   cqlrec_stream stream);
/* n_steps whole steps (steps step0 .. step0+n_steps-1) of a single-rank job (ctx->world == 1) in one call: the epoch
 * loop of TorchRecommender.train (replay/models/base_torch_rec.py:78-84) without a host round trip per batch.
 * Same dataflow and results as fwd_bwd + update per step, software-pipelined across the two halves of the model:
 * Adam on E_in + encoder runs under the (MFMA-bound) item-side backward, the next step's sample / window gathers /
 * encoder start while the item-side Adam is still running, and only the Q-head kernels wait for it.  Everything is
 * joined on `stream` before the call returns.  loss_out: n_steps device floats (may be NULL). */
int cqlrec_train_steps(const cqlrec_train_ctx* ctx /* [host] */, uint64_t step0, int32_t n_steps, float* loss_out,
                       cqlrec_stream stream);

/* 1 (default; env CQL_CONCURRENCY=0 to disable): independent parts of a step (the sort for the gather backward, the
 * two forward branches, the item-side vs state-side Q-head backward) run on internal side streams, forked and joined
 * with events on `stream`.  0: strict program order on `stream` -- use it when timing individual kernels. */
int cqlrec_set_concurrency(int32_t on);

/* The step driver's internal side streams and events of the CURRENT device, chosen and created NOW instead of at the
 * first training step, plus CQLREC_AUX_STREAMS library-owned streams for the caller's own side work:
 * cqlrec_aux_stream(0) = the data-parallel loop's item-side stream (the stream cqlrec_train_steps uses for its
 * sample-ahead, which a phased caller never runs), cqlrec_aux_stream(1) = the predict pass's encoder stream; NULL before
 * cqlrec_runtime_init or for a bad index.
 * WHY (measured on MI355X / ROCm 7.2: tools/probes/pipe_probe.hip, tools/stream_order_probe3.py): a process's hardware
 * queues sit on the 4 compute pipes round-robin in the order in which its streams were first used, and while a grid
 * larger than the chip is being handed out on one queue, no kernel of another queue ON THE SAME PIPE is dispatched.  The
 * step runs four streams side by side; if two of them share a pipe -- which used to depend on what else the process had
 * created streams for, and when -- the step loses its concurrency (cfg3: 1.33 instead of 0.69 ms per step).  So the
 * library picks its streams by test: out of up to eight fresh candidates, three that block neither the DEFAULT stream (the
 * stream the caller is expected to train on) nor each other.  One-time cost: a few milliseconds, a device
 * synchronisation; CQL_PIPE_PROBE=0 skips the test.  cqlrec_runtime_probe_count: candidates examined (0: no test ran).
 * Idempotent; needs a visible GPU; replay_cql_amd calls it when a model is constructed.
 * No reference counterpart (the reference trains on one stream, replay/models/base_torch_rec.py:57-98). */
#define CQLREC_AUX_STREAMS 2
int cqlrec_runtime_init(void);
int cqlrec_runtime_probe_count(void);
cqlrec_stream cqlrec_aux_stream(int32_t index);

/* Debug/inspection of the intermediates of step `step` inside ctx->ws (device pointers; valid after that step's
 * fwd_bwd, until step+2 overwrites them: the per-step vectors are double-buffered by step parity). */
typedef struct cqlrec_train_views {
  int32_t *users, *tpos, *act, *a_star;
  float *rew, *done, *q_a, *lse, *q_targ, *y, *coef, *dH, *dh0, *h0_s;
  uint16_t *hb_s, *hb_sn, *hb_tn;
} cqlrec_train_views;
int cqlrec_train_views_get(const cqlrec_train_ctx* ctx /* [host] */, uint64_t step,
                           cqlrec_train_views* out /* [host] */);

/* ---------------------------------------------------------------------------------------------------------
 * f2  log -> CSR by user on the device: rows sorted by (user_idx, timestamp asc, item_idx asc), S2.  Takes the place of
 * `log.toPandas()` + DataLoader construction (replay/models/neuromf.py:332-339).  Inputs are the LOG_SCHEMA columns
 * (replay/constants.py:16-23) as device arrays; timestamp in any monotone int64 unit.  Bit-exact vs a host lexsort.
 * timestamp == NULL: rows ordered by (user_idx, item_idx asc) instead -- the per-user ascending `seen` lists that
 * cqlrec_score_topk filters with (the role of the anti-join in _filter_seen, replay/models/base_rec.py:417-464).
 * relevance == NULL (then rewards must be NULL too): no reward column is produced.
 * --------------------------------------------------------------------------------------------------------- */
int64_t cqlrec_build_csr_ws_bytes(int64_t n_rows);
int cqlrec_build_csr(const int32_t* user_idx, const int32_t* item_idx, const int64_t* timestamp,
                     const double* relevance, int64_t n_rows, int64_t n_users, void* ws, int64_t ws_bytes,
                     int64_t* offsets /* [n_users+1] */, int32_t* items, float* rewards, cqlrec_stream stream);

/* ---------------------------------------------------------------------------------------------------------
 * f4  Quality metrics of a recommendation block on the device, so that optimize()-style loops
 * (replay/optuna_objective.py:80-111) need not ship U*k rows back through Spark.  Per-user formulas:
 * replay/metrics/{ndcg.py:50-59, hitrate.py:22-27, precision.py, recall.py, map.py, mrr.py}; user set and
 * "no recommendations -> empty prediction" as get_enriched_recommendations (replay/metrics/base_metric.py:102-140).
 *   rec_idx  [n_users x kmax] item ids, best first, unique per row, -1 padded
 *   rec_rows [n_users] row of the ground-truth CSR for each rec row (NULL: identity)
 *   gt_off / gt_items  ground-truth CSR, items ascending and unique per row
 *   ks       [host] n_ks ascending cut-offs (<= kmax, n_ks <= 8)
 * Output: sums[CQLREC_EVAL_METRICS][n_ks] = per-metric sums over the n_users rows (divide by the number of
 * ground-truth users); per_user[n_users][CQLREC_EVAL_METRICS][n_ks] optional.  Order: NDCG, HitRate, Precision,
 * Recall, MAP, MRR.  Double precision, deterministic reduction.
 * --------------------------------------------------------------------------------------------------------- */
#define CQLREC_EVAL_METRICS 6
int64_t cqlrec_eval_topk_ws_bytes(int64_t n_users, int32_t n_ks);
int cqlrec_eval_topk(const int32_t* rec_idx, int64_t n_users, int32_t kmax, const int32_t* rec_rows,
                     const int64_t* gt_off, const int32_t* gt_items, const int32_t* ks /* [host] */, int32_t n_ks,
                     void* ws, int64_t ws_bytes, double* per_user, double* sums, cqlrec_stream stream);

/* ---------------------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py): when enabled, every launcher brackets its kernel with a pair of HIP events on
 * the stream it launches on; cqlrec_prof_read synchronises those events and returns, per phase, the summed
 * kernel time in ms and the number of launches, then resets the pool.  Not capturable in a hipGraph; off by
 * default.  The reference's only harness is time.time() around fit/predict
 * (experiments/02_models_comparison.ipynb:843-867).
 * --------------------------------------------------------------------------------------------------------- */
enum {
  CQLREC_PH_SAMPLE = 0, CQLREC_PH_GATHER_FWD, CQLREC_PH_ENCODER_FWD, CQLREC_PH_QHEAD_LSE, CQLREC_PH_QHEAD_ARGMAX,
  CQLREC_PH_QHEAD_BWD_DH, CQLREC_PH_QHEAD_BWD_DE, CQLREC_PH_QHEAD_SMALL, CQLREC_PH_ENCODER_BWD,
  CQLREC_PH_GATHER_BWD, CQLREC_PH_ADAM, CQLREC_PH_TOPK_TILEMAX, CQLREC_PH_TOPK_SELECT, CQLREC_PH_COUNT
};
int cqlrec_prof_enable(int32_t on);
/* Restrict the bracketing to the phases whose bit (1u << CQLREC_PH_*) is set; default: all.  Every event pair
 * costs two barrier packets on its stream, so a throughput measurement brackets only the kernel it reports on. */
int cqlrec_prof_select(uint32_t phase_mask);
int cqlrec_prof_read(double* ms_sum /* [host] CQLREC_PH_COUNT */, int64_t* launches /* [host] CQLREC_PH_COUNT */);

/* Schedule marks (tools/phase_timing.py): nine timing events at the joints of the middle step of a
 * cqlrec_train_steps call (n_steps >= 4) -- loss, dH done, item-side backward done, encoder+gather backward done,
 * Adam(E_in) done, Adam(E_out) done, next step's prologue done, its LSE done, its loss; then encoder dx done, window-
 * gather backward done, the next step's sample + pair sorts done -- cheap enough not to disturb
 * the overlap they measure.  marks_read synchronises and returns ms relative to the first mark (-1: not recorded). */
#define CQLREC_DEBUG_MARKS 12
int cqlrec_debug_marks_enable(int32_t on);
int cqlrec_debug_marks_read(float* ms_out /* [host] CQLREC_DEBUG_MARKS */);

#ifdef __cplusplus
}
#endif
#endif /* CQLREC_H */
