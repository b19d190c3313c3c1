// Internal interface of the streaming Q-head skeleton (shared by qhead.hip and topk.hip).
#pragma once
#include "common.h"

#define QM_LSE 1
#define QM_ARGMAX 2
#define QM_TILEMAX 3
#define QM_BWD_DH 4
#define QM_BWD_DE 5
#define QM_LSE_DH 6         // forward logsumexp AND the softmax-weighted item sum (the soft part of dH) in one pass
#define QM_TOPK 7           // running top-k (k <= 16) per user in the epilogue: no score or group maximum leaves the chip
#define QM_TOPK10 8         // the same with only 10 of the 16 list entries kept sorted (k <= 10; see qtopk4_kernel's KC)
#define QS_TOPK_K 16        // list length kept per (user, lane half, item slice)
#define QS_TOPK_BUF 4       // candidates buffered per lane between two merges into the list

#define QS_TI 64            // host-side unit of streamed rows (split boundaries are multiples of it)
#define QS_SPW_FWD 2        // 32-row owner groups per wave, forward modes (64 states per wave, 256 per block)
#define QS_SPW_BWD 1        // backward modes (32 owners per wave, 128 per block)
#define QS_TARGET_BLOCKS 768      // forward modes: 3 blocks per CU resident
#define QS_TARGET_BLOCKS_BWD 512  // backward modes: 2 blocks per CU resident (register budget)
#define QS_REF_MARGIN 5.5f   // fused forward: the running reference jumps this far (nats) above a tile maximum that beat it
#ifndef QS_NBUF
#define QS_NBUF 3             // LDS stage buffers (prefetch distance NBUF-1 stages), forward modes
#endif
#ifndef QS_NBUF_BWD
#define QS_NBUF_BWD 2         // ... two-MFMA modes (BWD_DH, BWD_DE, LSE_DH)
#endif

struct QArgs {
  const uint16_t* res;       // owner rows   [n_res x D] bf16
  int64_t n_res;
  const uint16_t* str;       // streamed rows [n_str x D] bf16
  int64_t n_str;
  const float* str_scalar;   // per streamed row: bias (fwd, BWD_DH) / -lse*log2e (BWD_DE)
  const float* res_scalar;   // per owner row:    -lse*log2e (BWD_DH) / bias (BWD_DE)
  int nsplit;
  int64_t split_rows;        // streamed rows per slice (multiple of QS_TI and of 32*tg)
  float* part_a;             // [nsplit][n_res]  running max
  float* part_b;             // [nsplit][n_res]  running sum (LSE)
  int32_t* part_i;           // [nsplit][n_res]  argmax
  float* tilemax;            // [ngroups][n_res] (TILEMAX)
  int tg;                    // 32-row tiles per tile group (TILEMAX)
  float* slab;               // [nsplit][n_res][D] (backward)
  float* slab_cs;            // [nsplit][n_res]    (BWD_DE column sums of P)
  // direct output (backward, nsplit == 1): out[row][D] = scale * y, out_cs[row] = scale * colsum; no slab round trip
  float* out;
  float* out_cs;
  float scale;
  int accumulate;            // direct output adds to out / out_cs instead of overwriting (rows are block-owned: no atomics)
  // QM_TOPK: per (slice, user, lane half) the QS_TOPK_K best admissible candidates as sortable 64-bit keys
  // (order-preserving score bits << 32 | ~candidate row), best first, 0 = none
  const int* guard;                // QM_LSE_DH as a fall-back launch: every block returns at once unless *guard != 0
  unsigned long long* topk_keys;   // [nsplit][n_res][2][QS_TOPK_K]
  int topk_k;                      // requested k (<= QS_TOPK_K): the pruning threshold is the k-th best so far
  const uint32_t* seen_bits;       // [n_res][seen_w] bitmap of the rows to exclude per owner row (NULL: no filter)
  int64_t seen_w;                  // 32-bit words per owner row
};

struct QSplit {
  int nsplit;
  int64_t split_rows;
  int64_t rblks;
};

int qs_spw_fwd(int d);
QSplit qs_choose_split(int64_t n_str, int64_t n_res, int spw, int unit_rows, int target_blocks);
int qs_launch(int mode, const QArgs& a, int d, int64_t rblks, hipStream_t s);
// item-side backward for the step driver: the one-hot scatter goes FIRST (into zeroed g_E_out / g_b_out), the streaming
// kernel then adds its rows -- nothing small is left behind the long kernel on the step's critical path.
// Fused forward for training ("flash" form): one pass over the catalogue yields, per state, the logsumexp AND
// sum_j exp(S_j - m) E_out_b[j] relative to a running reference m (slabs in `ws`), so that the state-side backward
// needs no second pass over the catalogue -- cql_qhead_dh_finish turns the slabs into dH once lse and the TD
// coefficients are known:  dH[b] = scale * sum_k slab_k[b] * exp(m_k[b] - lse[b]) + coef[b] * E_out_b[act[b]].
// ws: cqlrec_qhead_bwd_ws_bytes(rows, n_items, d) + cqlrec_qhead_ws_bytes(rows, n_items, d) bytes.
int cql_qhead_fwd_lse_dh(const uint16_t* H_b, int64_t rows, const uint16_t* E_out_b, const float* b_out, int64_t n_items,
                         int32_t d, void* ws, int64_t ws_bytes, float* out_lse, float* out_nlse2, hipStream_t stream,
                         float* out_nlse_nat = nullptr,       // out_nlse_nat: -lse in natural units (what qde2 wants)
                         int flag_cleared = 0);               // cql_qhead_fwd_lse_dh_prepare ran on this stream already
int cql_qhead_fwd_lse_dh_prepare(void* ws, int64_t rows, int64_t n_items, int32_t d, hipStream_t s);
int cql_qhead_dh_finish(const void* ws, int64_t rows, int64_t n_items, int32_t d, const float* lse, const float* coef,
                        const int32_t* act, const uint16_t* E_out_b, float scale, float* dH, hipStream_t stream,
                        int part = 0);     // 0: all of dH; 1: the soft part (coef may be NULL); 2: + coef * E_out_b[a] (1 then 2 = 0, bit for bit)
// do_sparse: issue the scatter in this call; [item_lo, item_hi): item rows the streaming kernel handles in this call.
int cql_qhead_bwd_items_acc(const uint16_t* H_b, const float* nlse2, const float* coef, const int32_t* act, int64_t batch,
                            const uint16_t* E_out_b, const float* b_out, int64_t n_items, int32_t d, float scale, void* ws,
                            int64_t ws_bytes, float* g_E_out, float* g_b_out, hipStream_t stream, int do_sparse,
                            int64_t item_lo, int64_t item_hi, CqlAdamFix* defer = nullptr, const float* nlse_nat = nullptr);


// qhead_de.hip: the item-side backward as a persistent, statically balanced kernel (rows [0, n_items) of E_b / bias / out)
int64_t cql_qde_ws_bytes(int64_t batch, int64_t n_items, int32_t d);
int cql_qde_launch(const uint16_t* H_b, const float* nlse2, int64_t batch, const uint16_t* E_b, const float* bias,
                   int64_t n_items, int32_t d, float scale, void* ws, int64_t ws_bytes, float* out, float* out_cs,
                   int accumulate, hipStream_t s, CqlAdamFix* defer = nullptr, const float* nlse_nat = nullptr);

int cql_qde_fixup_deferred(const CqlAdamFix& f, float* out, float* out_cs, hipStream_t s);
// the long item-side kernel alone: rows WRITTEN (out = scale * dE, nothing read), no one-hot part; cut pieces left to
// `defer` (cql_qde_fixup_deferred / cql_adam_ema_fix) when the shape takes the stream-K kernels, else complete
int cql_qhead_bwd_items_long(const uint16_t* H_b, const float* nlse2, const float* coef, const int32_t* act, int64_t batch,
                             const uint16_t* E_out_b, const float* b_out, int64_t n_items, int32_t d, float scale, void* ws,
                             int64_t ws_bytes, float* g_E_out, float* g_b_out, hipStream_t stream, CqlAdamFix* defer,
                             const float* nlse_nat);

// qhead_topk2.hip: the top-K pass as a one-wave-per-SIMD kernel with on-chip selection (d = 128, k <= 16, whole catalogue)
struct QTk2Args {
  const uint16_t* H_b;          // [n_users x D] bf16 state vectors
  int64_t n_users;
  const uint16_t* E_b;          // [n_cand x D] bf16 item rows
  const float* bias;            // [n_cand]
  int64_t n_cand;
  int64_t split_rows;           // item rows per slice (multiple of 64)
  int nsplit;
  const uint32_t* seen_bits;    // cql_topk2_seen_bits layout, or NULL (no filter)
  unsigned long long* keys;     // [nsplit][n_users][2][QS_TOPK_K]
  int k;
  // qtopk4_kernel only: the seen filter as entry lists (cql_topk2_seen_bits builds them IN the bitmap's space when the
  // shape takes that kernel; the bitmap itself only when they do not fit) and the word that picks between the two forms
  const uint32_t* seen_lists;        // [ceil(n_users / 128)][stages][64 words] slots, or NULL
  const uint16_t* seen_lists_ovf;    // overflow entries
  const uint32_t* guard;             // NULL: run; else run iff (*guard != 0) == (guard_want != 0)
  int guard_want;
};
bool cql_topk2_supported(int d, int k, int64_t n_cand);
void cql_topk2_split(int64_t n_users, int64_t n_cand, int* nsplit, int64_t* split_rows);
int64_t cql_topk2_bits_bytes(int64_t n_users, int64_t n_cand);
const uint32_t* cql_topk2_lists_word(const uint32_t* bits, int64_t n_users, int64_t n_cand);    // NULL: bitmap only
int cql_topk2_seen_bits(const int64_t* seen_off, const int32_t* seen_items, const int32_t* seen_rows, int64_t n_users,
                        int64_t n_cand, uint32_t* bits, hipStream_t s, int beside_scoring = 0);
int cql_topk2_run(const QTk2Args& a, int d, hipStream_t s);
// qhead_topk4.hip: the same pass with four user groups per wave (512 users per block); chosen by shape inside
// cql_topk2_split / cql_topk2_run, so the callers of those two need not know
bool cql_topk4_use(int d, int k, int64_t n_users, int64_t n_cand);
int cql_topk4_run(const QTk2Args& a, hipStream_t s);
bool cql_topk4_lists_on();
bool cql_topk4_lists_fit(int64_t n_users, int64_t n_cand, int64_t space_bytes);
int cql_topk4_seen_lists(const int64_t* seen_off, const int32_t* seen_items, const int32_t* seen_rows, int64_t n_users,
                         int64_t n_cand, void* space, int64_t space_bytes, uint32_t* flag, hipStream_t s);

int cql_qhead_argmax_beside(const uint16_t* H_b, int64_t rows, const uint16_t* E_out_b, const float* b_out, int64_t n_items,
                            int32_t d, void* ws, int64_t ws_bytes, float* out_val, int32_t* out_idx, hipStream_t stream);

int cql_qhead_argmax_step(const uint16_t* H_b, int64_t rows, const uint16_t* E_out_b, const float* b_out, int64_t n_items,
                          int32_t d, void* ws, int64_t ws_bytes, float* out_val, int32_t* out_idx, hipStream_t stream);

// qhead_fwd2.hip: the fused forward (lse + softmax-weighted item sum) as a one-wave-per-SIMD kernel (d = 128)
struct QFwd2Args {
  const uint16_t* H_b;      // [n_states x D] owner rows
  int64_t n_states;
  const uint16_t* E_b;      // [n_items x D] streamed rows
  const float* bias;        // [n_items]
  int64_t n_items;
  int nsplit;
  int64_t split_rows;       // items per slice (multiple of 64)
  float* slab;              // [nsplit][n_states][D]
  float* part_a;            // [nsplit][n_states] reference (natural units)
  float* part_b;            // [nsplit][n_states] sum of exp(S - reference)
  int* flag;                // set when a partial sum is not finite (the guarded first form then redoes the pass)
};
bool cql_qfwd2_supported(int d, int64_t n_items);
int cql_qfwd2_run(const QFwd2Args& a, int d, hipStream_t s);
// qhead_argmax2.hip: the ARGMAX pass as a one-wave-per-SIMD kernel (d = 128, 256); partials in QM_ARGMAX's format
bool cql_qargmax2_supported(int d, int64_t n_items);
void cql_qargmax2_split(int64_t rows, int64_t n_items, int d, int* nsplit, int64_t* split_rows);
int cql_qargmax2_run(const uint16_t* H_b, int64_t rows, const uint16_t* E_b, const float* bias, int64_t n_items, int d,
                     int nsplit, int64_t split_rows, float* part_v, int32_t* part_i, hipStream_t s);
// qhead_fwd3.hip: the same pass for d = 256 (one 32-state group per wave, 128 states per block)
bool cql_qfwd3_supported(int d, int64_t n_items);
int cql_qfwd3_run(const QFwd2Args& a, int d, hipStream_t s);
