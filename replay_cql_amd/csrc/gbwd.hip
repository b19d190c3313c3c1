// Backward of the window gather (a6 tail): g_E_in[item] += dh0[b] / len_b for every item of every window.
//
// A plain scatter of B*L rows with fp32 atomics runs at the chip-wide atomic rate and, on Zipf-distributed logs,
// hammers a few hot rows (MI355X_MICROARCH.md "Global float atomics": every adder on ONE row = 14x slower).
// Instead (cdna_hip_programming.md Appendix B "Scatter / gather / embedding"):
//   1. one wave per state writes its scaled row g[b] = dh0[b]/len_b and its <= L (item, b) pairs (pad key = N);
//   2. rocPRIM radix sort of the pairs by item (stable: contributions of one item stay in state order);
//   3. segmented sum, deterministic (no atomics): every wave walks 64 consecutive sorted pairs with the running row
//      sum in registers and stores once per run; pieces of runs that cross a chunk edge are combined by a second
//      small pass in chunk order (hot items: one add per 64 contributions, then one wave sums the pieces).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

__global__ __launch_bounds__(256) void gbwd_pairs_kernel(const int64_t* __restrict__ offsets,
                                                         const int32_t* __restrict__ items,
                                                         const int32_t* __restrict__ users,
                                                         const int32_t* __restrict__ ends, int end_delta,
                                                         int64_t n_states, int L, uint32_t pad_key,
                                                         uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                         int32_t* __restrict__ lens) {
  const int lane = threadIdx.x & 63;
  const int64_t state = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (state >= n_states) return;
  const int u = users[state];
  const int64_t o0 = offsets[u];
  const int end = ends ? (ends[state] + end_delta) : (int)(offsets[u + 1] - o0);
  const int len = end < L ? end : L;
  if (lane == 0) lens[state] = len;
  const int32_t* win = items + o0 + end - len;
  for (int j = lane; j < L; j += 64) {
    keys[state * L + j] = (j < len) ? (uint32_t)win[j] : pad_key;
    vals[state * L + j] = (uint32_t)state;
  }
}

// g[b][:] = dh0[b][:] / len_b  (one IEEE division per element, as the oracle does)
__global__ __launch_bounds__(256) void gbwd_scale_kernel(const float4* __restrict__ dh0, const int32_t* __restrict__ lens,
                                                         int64_t n4, int d4, float4* __restrict__ g) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int len = lens[i / d4];
  const float fl = (float)len;
  const float4 v = dh0[i];
  g[i] = len > 0 ? make_float4(v.x / fl, v.y / fl, v.z / fl, v.w / fl) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// Segmented sum over sorted (key, row index) pairs, DETERMINISTIC: dst[key][:] = sum over the key's run, in sorted
// order, of src[row][:] (fp32 rows; or bf16 rows times a per-row fp32 weight, SRC_BF16).  dst rows must be zero (or
// untouched by anyone else) on entry: every row is written exactly once, by a plain store.
//   pass 1: every wave walks 64 consecutive pairs with the running row sum in registers.  A run that lies inside the
//           chunk is stored to its row; the piece of a run that crosses a chunk edge goes to edge[chunk][0] (run came
//           in from the left) or edge[chunk][1] (run leaves to the right, started here).
//   pass 2: the wave of the chunk in which a crossing run STARTS adds the pieces in chunk order and stores the row.
// (A hot Zipf item's run spans hundreds of chunks: one add per 64 contributions, then one wave sums the pieces.)
// No float atomics anywhere: the result is bit-reproducible from run to run and independent of scheduling.
template <int D, bool SRC_BF16>
__device__ __forceinline__ void segsum_add_row(float (&acc)[D / 64], float& accw, const void* __restrict__ src,
                                               const float* __restrict__ w, uint32_t row, int lane) {
  constexpr int PER = D / 64;
  if constexpr (SRC_BF16) {
    const float wt = w[row];
    const uint16_t* h = (const uint16_t*)src + (int64_t)row * D;
#pragma unroll
    for (int k = 0; k < PER; ++k) acc[k] += wt * bf16_bits_to_f32(h[k * 64 + lane]);
    accw += wt;
  } else {
    const float* g = (const float*)src + (int64_t)row * D;
#pragma unroll
    for (int k = 0; k < PER; ++k) acc[k] += g[k * 64 + lane];
  }
}

// CH = pairs per wave (64 for the window gather: long runs of hot items; 8 for the one-hot scatter: only B pairs, so
// small chunks are what gives the launch enough waves).
template <int D, bool SRC_BF16, int CH>
__global__ __launch_bounds__(256) void segsum_pass1_kernel(const void* __restrict__ src, const float* __restrict__ w,
                                                           const uint32_t* __restrict__ keys,
                                                           const uint32_t* __restrict__ vals, int64_t n_pairs,
                                                           uint32_t pad_key, float* __restrict__ dst,
                                                           float* __restrict__ dst_w, float* __restrict__ edge,
                                                           float* __restrict__ edge_w, int accumulate) {
  constexpr int PER = D / 64;
  const int lane = threadIdx.x & 63;
  const int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t chunk0 = chunk * CH;
  if (chunk0 >= n_pairs) return;
  const int64_t idx = chunk0 + lane;
  const bool mine = lane < CH && idx < n_pairs;
  const uint32_t my_key = mine ? keys[idx] : pad_key;
  const uint32_t my_val = mine ? vals[idx] : 0u;
  if (__builtin_amdgcn_readfirstlane(my_key) == pad_key) return;   // sorted: the whole chunk is padding
  const uint32_t prev_key = (chunk0 > 0) ? keys[chunk0 - 1] : pad_key;
  const uint32_t next_key = (chunk0 + CH < n_pairs) ? keys[chunk0 + CH] : pad_key;
  float acc[PER], accw = 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k) acc[k] = 0.f;
  uint32_t run_key = __builtin_amdgcn_readfirstlane(my_key);
  bool open_left = (run_key == prev_key);
  auto flush = [&](bool open_right) {
    float* row;
    float* roww = nullptr;
    if (open_left) { row = edge + (chunk * 2 + 0) * D; if (SRC_BF16) roww = edge_w + chunk * 2 + 0; }
    else if (open_right) { row = edge + (chunk * 2 + 1) * D; if (SRC_BF16) roww = edge_w + chunk * 2 + 1; }
    else { row = dst + (int64_t)run_key * D; if (SRC_BF16) roww = dst_w + run_key; }
    const bool add = accumulate && row != edge + (chunk * 2 + 0) * D && row != edge + (chunk * 2 + 1) * D;   // rows of dst only
#pragma unroll
    for (int k = 0; k < PER; ++k) row[k * 64 + lane] = add ? row[k * 64 + lane] + acc[k] : acc[k];
    if (SRC_BF16 && lane == 0) *roww = add ? *roww + accw : accw;
  };
#pragma unroll 8
  for (int j = 0; j < CH; ++j) {
    const uint32_t kj = __shfl(my_key, j);
    if (kj != run_key) {   // the run ended inside the chunk (wave-uniform branch)
      flush(false);
#pragma unroll
      for (int k = 0; k < PER; ++k) acc[k] = 0.f;
      accw = 0.f;
      run_key = kj;
      open_left = false;
      if (kj == pad_key) return;
    }
    segsum_add_row<D, SRC_BF16>(acc, accw, src, w, __shfl(my_val, j), lane);
  }
  flush(run_key == next_key);
}

template <int D, bool SRC_BF16, int CH>
__global__ __launch_bounds__(256) void segsum_pass2_kernel(const uint32_t* __restrict__ keys, int64_t n_pairs,
                                                           uint32_t pad_key, const float* __restrict__ edge,
                                                           const float* __restrict__ edge_w, float* __restrict__ dst,
                                                           float* __restrict__ dst_w, int accumulate) {
  constexpr int PER = D / 64;
  const int lane = threadIdx.x & 63;
  const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n_chunks = (n_pairs + CH - 1) / CH;
  if (c + 1 >= n_chunks) return;
  auto key_at = [&](int64_t i) { return i < n_pairs ? keys[i] : pad_key; };
  const uint32_t k = key_at(c * CH + CH - 1);
  if (k == pad_key || key_at((c + 1) * CH) != k) return;          // no run leaves this chunk to the right
  if (key_at(c * CH) == k && c > 0 && key_at(c * CH - 1) == k) return;   // the run only passes through: not its start
  // last chunk that holds a piece of the run: the sorted keys make it a binary search for the end of the run
  int64_t lo = (c + 1) * CH, hi = n_pairs;     // first pair index >= lo whose key differs from k lies in (lo, hi]
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] == k) lo = mid + 1; else hi = mid;
  }
  const int64_t j_end = (lo - 1) / CH;          // chunk of the run's last pair
  float acc[PER], accw = 0.f;
#pragma unroll
  for (int q = 0; q < PER; ++q) acc[q] = edge[(c * 2 + 1) * D + q * 64 + lane];
  if (SRC_BF16) accw = edge_w[c * 2 + 1];
  // fixed order (chunk order), but with a known trip count the loads of several pieces are in flight together
#pragma unroll 4
  for (int64_t j = c + 1; j <= j_end; ++j) {
#pragma unroll
    for (int q = 0; q < PER; ++q) acc[q] += edge[(j * 2 + 0) * D + q * 64 + lane];
    if (SRC_BF16) accw += edge_w[j * 2 + 0];
  }
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    float* o = dst + (int64_t)k * D + q * 64 + lane;
    *o = accumulate ? *o + acc[q] : acc[q];
  }
  if (SRC_BF16 && lane == 0) dst_w[k] = accumulate ? dst_w[k] + accw : accw;
}

// host side of the two passes
template <bool SRC_BF16, int CH>
static void segsum_launch(const void* src, const float* w, const uint32_t* keys, const uint32_t* vals, int64_t n_pairs,
                          uint32_t pad_key, int d, float* dst, float* dst_w, float* edge, float* edge_w, hipStream_t s,
                          int accumulate = 0) {
  dim3 grid(cql_ceil_div((n_pairs + CH - 1) / CH, 4)), block(256);
#define SS1(DD) hipLaunchKernelGGL((segsum_pass1_kernel<DD, SRC_BF16, CH>), grid, block, 0, s, src, w, keys, vals, n_pairs, pad_key, dst, dst_w, edge, edge_w, accumulate)
#define SS2(DD) hipLaunchKernelGGL((segsum_pass2_kernel<DD, SRC_BF16, CH>), grid, block, 0, s, keys, n_pairs, pad_key, edge, edge_w, dst, dst_w, accumulate)
  if (d == 64) { SS1(64); SS2(64); } else if (d == 128) { SS1(128); SS2(128); } else { SS1(256); SS2(256); }
#undef SS1
#undef SS2
}

#define GB_CH 64   // pairs per wave, window gather
#define OH_CH 8    // pairs per wave, one-hot scatter

static inline int64_t a256(int64_t x) { return (x + 255) / 256 * 256; }
static inline unsigned key_bits(int64_t n_items) {
  unsigned b = 1;
  while ((1ll << b) <= n_items) ++b;   // pad key = n_items must be representable
  return b;
}
static int64_t sort_temp_bound(int64_t n) { return a256(2 * n * 4 * 2) + (8ll << 20); }

namespace {
struct GbWs {
  float* g;
  float* edge;      // [n_chunks][2][d] pieces of runs that cross a chunk edge
  int32_t* lens;
  uint32_t *keys_in, *vals_in, *keys_out, *vals_out;
  void* temp;
  size_t temp_cap;
  int64_t total;
};
GbWs gb_carve(void* ws, int64_t n_states, int32_t L, int32_t d) {
  const int64_t n = n_states * L;
  char* p = (char*)ws;
  GbWs w;
  w.g = (float*)p;              p += a256(n_states * d * 4);
  w.edge = (float*)p;           p += a256(((n + GB_CH - 1) / GB_CH) * 2 * d * 4);
  w.lens = (int32_t*)p;         p += a256(n_states * 4);
  w.keys_in = (uint32_t*)p;     p += a256(n * 4);
  w.vals_in = (uint32_t*)p;     p += a256(n * 4);
  w.keys_out = (uint32_t*)p;    p += a256(n * 4);
  w.vals_out = (uint32_t*)p;    p += a256(n * 4);
  w.temp = p;
  w.temp_cap = (size_t)sort_temp_bound(n);
  w.total = (int64_t)(p - (char*)ws) + (int64_t)w.temp_cap;
  return w;
}
}  // namespace

extern "C" int64_t cqlrec_gather_pool_bwd_ws_bytes(int64_t n_states, int32_t L, int32_t d) {
  return gb_carve(nullptr, n_states, L, d).total + 256;
}

// phase 1 (needs only the sampled states, not the gradient): window pairs + radix sort by item.  Independent of
// the forward/backward math, so the step driver runs it on a side stream underneath the Q-head kernels.
extern "C" int cqlrec_gather_pool_bwd_prepare(const int64_t* offsets, const int32_t* items, const int32_t* users,
                                              const int32_t* ends, int32_t end_delta, int64_t n_states, int32_t L,
                                              int32_t d, int64_t n_items, void* ws, int64_t ws_bytes,
                                              cqlrec_stream stream) {
  CQL_REQUIRE(offsets && items && users && ws, "gather_pool_bwd_prepare: NULL pointer");
  CQL_REQUIRE(d == 64 || d == 128 || d == 256, "gather_pool_bwd_prepare: d=%d unsupported", d);
  CQL_REQUIRE(n_items > 0 && n_items < (1ll << 31), "gather_pool_bwd_prepare: n_items=%lld", (long long)n_items);
  if (n_states <= 0) return CQLREC_OK;
  CQL_REQUIRE(ws_bytes >= cqlrec_gather_pool_bwd_ws_bytes(n_states, L, d), "gather_pool_bwd_prepare: workspace too small");
  const GbWs w = gb_carve(ws, n_states, L, d);
  const int64_t n = n_states * L;
  hipStream_t s = (hipStream_t)stream;
  const uint32_t pad_key = (uint32_t)n_items;
  CqlProfScope prof(CQLREC_PH_GATHER_BWD, s);
  hipLaunchKernelGGL(gbwd_pairs_kernel, dim3(cql_ceil_div(n_states, 4)), dim3(256), 0, s, offsets, items, users, ends,
                     end_delta, n_states, L, pad_key, w.keys_in, w.vals_in, w.lens);
  size_t need = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, need, w.keys_in, w.keys_out, w.vals_in, w.vals_out, (size_t)n, 0u,
                                           key_bits(n_items), s);
  if (e != hipSuccess || need > w.temp_cap) {
    cql_set_error("gather_pool_bwd_prepare: radix sort needs %zu bytes of scratch (have %zu), err=%d", need, w.temp_cap,
                  (int)e);
    return CQLREC_ERR_HIP;
  }
  e = rocprim::radix_sort_pairs(w.temp, need, w.keys_in, w.keys_out, w.vals_in, w.vals_out, (size_t)n, 0u,
                                key_bits(n_items), s);
  if (e != hipSuccess) {
    cql_set_error("gather_pool_bwd_prepare: radix sort failed: %s", hipGetErrorString(e));
    return CQLREC_ERR_HIP;
  }
  CQL_LAUNCH_CHECK("gather_pool_bwd_prepare");
  return CQLREC_OK;
}

// phase 2: scale the gradient rows and sum the sorted runs into g_E_in (must be zero on entry)
extern "C" int cqlrec_gather_pool_bwd_apply(const float* dh0, int64_t n_states, int32_t L, int32_t d, int64_t n_items,
                                            void* ws, int64_t ws_bytes, float* g_E_in, cqlrec_stream stream) {
  CQL_REQUIRE(dh0 && ws && g_E_in, "gather_pool_bwd_apply: NULL pointer");
  CQL_REQUIRE(d == 64 || d == 128 || d == 256, "gather_pool_bwd_apply: d=%d unsupported", d);
  if (n_states <= 0) return CQLREC_OK;
  CQL_REQUIRE(ws_bytes >= cqlrec_gather_pool_bwd_ws_bytes(n_states, L, d), "gather_pool_bwd_apply: workspace too small");
  const GbWs w = gb_carve(ws, n_states, L, d);
  const int64_t n = n_states * L;
  hipStream_t s = (hipStream_t)stream;
  const uint32_t pad_key = (uint32_t)n_items;
  CqlProfScope prof(CQLREC_PH_GATHER_BWD, s);
  const int64_t n4 = n_states * (d / 4);
  hipLaunchKernelGGL(gbwd_scale_kernel, dim3(cql_ceil_div(n4, 256)), dim3(256), 0, s, (const float4*)dh0, w.lens, n4,
                     d / 4, (float4*)w.g);
  segsum_launch<false, GB_CH>(w.g, nullptr, w.keys_out, w.vals_out, n, pad_key, d, g_E_in, nullptr, w.edge, nullptr, s);
  CQL_LAUNCH_CHECK("gather_pool_bwd_apply");
  return CQLREC_OK;
}

extern "C" int cqlrec_gather_pool_bwd_sorted(const float* dh0, const int64_t* offsets, const int32_t* items,
                                             const int32_t* users, const int32_t* ends, int32_t end_delta,
                                             int64_t n_states, int32_t L, int32_t d, int64_t n_items, void* ws,
                                             int64_t ws_bytes, float* g_E_in, cqlrec_stream stream) {
  int rc = cqlrec_gather_pool_bwd_prepare(offsets, items, users, ends, end_delta, n_states, L, d, n_items, ws, ws_bytes,
                                          stream);
  if (rc != CQLREC_OK) return rc;
  return cqlrec_gather_pool_bwd_apply(dh0, n_states, L, d, n_items, ws, ws_bytes, g_E_in, stream);
}


// =============================================================================================================
// One-hot part of the Q-head backward, deterministic:  g_E_out[a] += sum_{b: act[b]=a} coef[b] H_b[b],
// g_b_out[a] += sum coef[b]  -- the same sorted segmented sum, over the B pairs (act[b], b).  The sort depends on
// the sampled actions only, so the step driver runs it ahead of time (with the pairs of the window gather).
// =============================================================================================================
namespace {
struct OhWs {
  uint32_t *keys_in, *vals_in, *keys_out, *vals_out;
  float *edge, *edge_w;
  void* temp;
  size_t temp_cap;
  int64_t total;
};
OhWs oh_carve(void* ws, int64_t batch, int32_t d) {
  char* p = (char*)ws;
  OhWs w;
  const int64_t n_chunks = (batch + OH_CH - 1) / OH_CH;
  w.keys_in = (uint32_t*)p;     p += a256(batch * 4);
  w.vals_in = (uint32_t*)p;     p += a256(batch * 4);
  w.keys_out = (uint32_t*)p;    p += a256(batch * 4);
  w.vals_out = (uint32_t*)p;    p += a256(batch * 4);
  w.edge = (float*)p;           p += a256(n_chunks * 2 * d * 4);
  w.edge_w = (float*)p;         p += a256(n_chunks * 2 * 4);
  w.temp = p;
  w.temp_cap = (size_t)sort_temp_bound(batch);
  w.total = (int64_t)(p - (char*)ws) + (int64_t)w.temp_cap;
  return w;
}
__global__ void onehot_pairs_kernel(const int32_t* __restrict__ act, int64_t n, uint32_t* __restrict__ keys,
                                    uint32_t* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = (uint32_t)act[i];
  vals[i] = (uint32_t)i;
}
}  // namespace

int64_t cql_onehot_ws_bytes(int64_t batch, int32_t d) { return oh_carve(nullptr, batch, d).total + 256; }

int cql_onehot_prepare(const int32_t* act, int64_t batch, int64_t n_items, int32_t d, void* ws, int64_t ws_bytes,
                       hipStream_t s) {
  CQL_REQUIRE(act && ws, "onehot_prepare: NULL pointer");
  CQL_REQUIRE(batch > 0 && ws_bytes >= cql_onehot_ws_bytes(batch, d), "onehot_prepare: workspace too small");
  const OhWs w = oh_carve(ws, batch, d);
  CqlProfScope prof(CQLREC_PH_GATHER_BWD, s);
  hipLaunchKernelGGL(onehot_pairs_kernel, dim3(cql_ceil_div(batch, 256)), dim3(256), 0, s, act, batch, w.keys_in, w.vals_in);
  size_t need = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, need, w.keys_in, w.keys_out, w.vals_in, w.vals_out, (size_t)batch, 0u,
                                           key_bits(n_items), s);
  if (e != hipSuccess || need > w.temp_cap) {
    cql_set_error("onehot_prepare: radix sort needs %zu bytes of scratch (have %zu), err=%d", need, w.temp_cap, (int)e);
    return CQLREC_ERR_HIP;
  }
  e = rocprim::radix_sort_pairs(w.temp, need, w.keys_in, w.keys_out, w.vals_in, w.vals_out, (size_t)batch, 0u,
                                key_bits(n_items), s);
  if (e != hipSuccess) {
    cql_set_error("onehot_prepare: radix sort failed: %s", hipGetErrorString(e));
    return CQLREC_ERR_HIP;
  }
  CQL_LAUNCH_CHECK("onehot_prepare");
  return CQLREC_OK;
}

// accumulate = 0: g_E_out / g_b_out rows of the sampled actions must be zero on entry (they are written, not added to);
// accumulate = 1: the sum of an action's run is ADDED to what its row holds (row + sum, once per row)
int cql_onehot_apply(const float* coef, const uint16_t* H_b, int64_t batch, int64_t n_items, int32_t d, void* ws,
                     float* g_E_out, float* g_b_out, hipStream_t s, int accumulate) {
  CQL_REQUIRE(coef && H_b && ws && g_E_out && g_b_out, "onehot_apply: NULL pointer");
  const OhWs w = oh_carve(ws, batch, d);
  CqlProfScope prof(CQLREC_PH_QHEAD_SMALL, s);
  segsum_launch<true, OH_CH>(H_b, coef, w.keys_out, w.vals_out, batch, (uint32_t)n_items, d, g_E_out, g_b_out, w.edge, w.edge_w, s, accumulate);
  CQL_LAUNCH_CHECK("onehot_apply");
  return CQLREC_OK;
}
