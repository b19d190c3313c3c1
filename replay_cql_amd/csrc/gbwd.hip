// Backward of the window gather (a6 tail): g_E_in[item] += dh0[b] / len_b for every item of every window.
//
// A plain scatter of B*L rows with fp32 atomics runs at the chip-wide atomic rate and, on Zipf-distributed logs,
// hammers a few hot rows (MI355X_MICROARCH.md "Global float atomics": every adder on ONE row = 14x slower).
// Instead (cdna_hip_programming.md Appendix B "Scatter / gather / embedding"):
//   1. one wave per state writes its scaled row g[b] = dh0[b]/len_b and its <= L (item, b) pairs (pad key = N);
//   2. rocPRIM radix sort of the pairs by item (stable: contributions of one item stay in state order);
//   3. segmented sum: every wave walks 64 consecutive sorted pairs, keeps the running row sum in registers and
//      flushes once per run -- a plain store when the run lies inside its chunk, one atomic row-add when the run
//      touches a chunk edge (hot items: one add per 64 contributions instead of 64).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

template <int D>
__global__ __launch_bounds__(256) void gbwd_pairs_kernel(const float* __restrict__ dh0, const int64_t* __restrict__ offsets,
                                                         const int32_t* __restrict__ items,
                                                         const int32_t* __restrict__ users,
                                                         const int32_t* __restrict__ ends, int end_delta,
                                                         int64_t n_states, int L, uint32_t pad_key,
                                                         float* __restrict__ g, uint32_t* __restrict__ keys,
                                                         uint32_t* __restrict__ vals) {
  const int lane = threadIdx.x & 63;
  const int64_t state = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (state >= n_states) return;
  const int u = users[state];
  const int64_t o0 = offsets[u];
  const int end = ends ? (ends[state] + end_delta) : (int)(offsets[u + 1] - o0);
  const int len = end < L ? end : L;
  const float fl = (float)len;
#pragma unroll
  for (int k = 0; k < D / 64; ++k) {
    const float v = dh0[state * D + k * 64 + lane];
    g[state * D + k * 64 + lane] = len > 0 ? v / fl : 0.f;
  }
  const int32_t* win = items + o0 + end - len;
  for (int j = lane; j < L; j += 64) {
    keys[state * L + j] = (j < len) ? (uint32_t)win[j] : pad_key;
    vals[state * L + j] = (uint32_t)state;
  }
}

template <int D>
__global__ __launch_bounds__(256) void gbwd_segsum_kernel(const float* __restrict__ g, const uint32_t* __restrict__ keys,
                                                          const uint32_t* __restrict__ vals, int64_t n_pairs,
                                                          uint32_t pad_key, float* __restrict__ g_E_in) {
  constexpr int PER = D / 64;
  const int lane = threadIdx.x & 63;
  const int64_t chunk0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  if (chunk0 >= n_pairs) return;
  const int64_t idx = chunk0 + lane;
  const uint32_t my_key = idx < n_pairs ? keys[idx] : pad_key;
  const uint32_t my_val = idx < n_pairs ? vals[idx] : 0u;
  if (__builtin_amdgcn_readfirstlane(my_key) == pad_key) return;   // sorted: the whole chunk is padding
  // does the first / last run of this chunk continue into the neighbouring chunks?
  const uint32_t prev_key = (chunk0 > 0) ? keys[chunk0 - 1] : pad_key;
  const uint32_t next_key = (chunk0 + 64 < n_pairs) ? keys[chunk0 + 64] : pad_key;
  float acc[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) acc[k] = 0.f;
  uint32_t run_key = __builtin_amdgcn_readfirstlane(my_key);
  bool run_open_left = (run_key == prev_key);
  for (int j = 0; j < 64; ++j) {
    const uint32_t kj = __shfl(my_key, j);
    if (kj != run_key) {   // flush the finished run (wave-uniform branch)
      float* row = g_E_in + (int64_t)run_key * D;
      if (run_open_left) {
#pragma unroll
        for (int k = 0; k < PER; ++k) atomicAdd(row + k * 64 + lane, acc[k]);
      } else {
#pragma unroll
        for (int k = 0; k < PER; ++k) row[k * 64 + lane] = acc[k];
      }
#pragma unroll
      for (int k = 0; k < PER; ++k) acc[k] = 0.f;
      run_key = kj;
      run_open_left = false;
      if (kj == pad_key) return;
    }
    const uint32_t bj = __shfl(my_val, j);
#pragma unroll
    for (int k = 0; k < PER; ++k) acc[k] += g[(int64_t)bj * D + k * 64 + lane];
  }
  float* row = g_E_in + (int64_t)run_key * D;
  if (run_open_left || run_key == next_key) {
#pragma unroll
    for (int k = 0; k < PER; ++k) atomicAdd(row + k * 64 + lane, acc[k]);
  } else {
#pragma unroll
    for (int k = 0; k < PER; ++k) row[k * 64 + lane] = acc[k];
  }
}

static inline int64_t a256(int64_t x) { return (x + 255) / 256 * 256; }
static inline unsigned key_bits(int64_t n_items) {
  unsigned b = 1;
  while ((1ll << b) <= n_items) ++b;   // pad key = n_items must be representable
  return b;
}
static int64_t sort_temp_bound(int64_t n) { return a256(2 * n * 4 * 2) + (8ll << 20); }

extern "C" int64_t cqlrec_gather_pool_bwd_ws_bytes(int64_t n_states, int32_t L, int32_t d) {
  const int64_t n = n_states * L;
  return a256(n_states * d * 4) + 4 * a256(n * 4) + sort_temp_bound(n) + 256;
}

extern "C" int cqlrec_gather_pool_bwd_sorted(const float* dh0, const int64_t* offsets, const int32_t* items,
                                             const int32_t* users, const int32_t* ends, int32_t end_delta,
                                             int64_t n_states, int32_t L, int32_t d, int64_t n_items, void* ws,
                                             int64_t ws_bytes, float* g_E_in, cqlrec_stream stream) {
  CQL_REQUIRE(dh0 && offsets && items && users && ws && g_E_in, "gather_pool_bwd_sorted: NULL pointer");
  CQL_REQUIRE(d == 64 || d == 128 || d == 256, "gather_pool_bwd_sorted: d=%d unsupported", d);
  CQL_REQUIRE(n_items > 0 && n_items < (1ll << 31), "gather_pool_bwd_sorted: n_items=%lld", (long long)n_items);
  if (n_states <= 0) return CQLREC_OK;
  CQL_REQUIRE(ws_bytes >= cqlrec_gather_pool_bwd_ws_bytes(n_states, L, d), "gather_pool_bwd_sorted: workspace too small");
  const int64_t n = n_states * L;
  char* p = (char*)ws;
  float* g = (float*)p;              p += a256(n_states * d * 4);
  uint32_t* keys_in = (uint32_t*)p;  p += a256(n * 4);
  uint32_t* vals_in = (uint32_t*)p;  p += a256(n * 4);
  uint32_t* keys_out = (uint32_t*)p; p += a256(n * 4);
  uint32_t* vals_out = (uint32_t*)p; p += a256(n * 4);
  void* temp = p;
  const size_t temp_cap = (size_t)sort_temp_bound(n);
  hipStream_t s = (hipStream_t)stream;
  const uint32_t pad_key = (uint32_t)n_items;
  CqlProfScope prof(CQLREC_PH_GATHER_BWD, s);
  dim3 grid(cql_ceil_div(n_states, 4)), block(256);
#define GBP(DD)                                                                                                      \
  hipLaunchKernelGGL(gbwd_pairs_kernel<DD>, grid, block, 0, s, dh0, offsets, items, users, ends, end_delta, n_states, L, \
                     pad_key, g, keys_in, vals_in)
  if (d == 64) GBP(64); else if (d == 128) GBP(128); else GBP(256);
#undef GBP
  size_t need = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, need, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u,
                                           key_bits(n_items), s);
  if (e != hipSuccess || need > temp_cap) {
    cql_set_error("gather_pool_bwd_sorted: radix sort needs %zu bytes of scratch (have %zu), err=%d", need, temp_cap, (int)e);
    return CQLREC_ERR_HIP;
  }
  e = rocprim::radix_sort_pairs(temp, need, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, key_bits(n_items), s);
  if (e != hipSuccess) {
    cql_set_error("gather_pool_bwd_sorted: radix sort failed: %s", hipGetErrorString(e));
    return CQLREC_ERR_HIP;
  }
  dim3 g2(cql_ceil_div(n, 256));
#define GBS(DD) hipLaunchKernelGGL(gbwd_segsum_kernel<DD>, g2, block, 0, s, g, keys_out, vals_out, n, pad_key, g_E_in)
  if (d == 64) GBS(64); else if (d == 128) GBS(128); else GBS(256);
#undef GBS
  CQL_LAUNCH_CHECK("gather_pool_bwd_sorted");
  return CQLREC_OK;
}
