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

namespace {
struct GbWs {
  float* g;
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
  dim3 g2(cql_ceil_div(n, 256)), block(256);
#define GBS(DD) hipLaunchKernelGGL(gbwd_segsum_kernel<DD>, g2, block, 0, s, w.g, w.keys_out, w.vals_out, n, pad_key, g_E_in)
  if (d == 64) GBS(64); else if (d == 128) GBS(128); else GBS(256);
#undef GBS
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
