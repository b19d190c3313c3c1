// Callers either side of the hot path (SURVEY 8(f) "next" rows), on the device:
//   f2  log -> CSR by user (sort by user, timestamp, item_idx) -- what NeuroMF._fit does with toPandas() + DataLoader
//       (replay/models/neuromf.py:332-339), here three stable rocPRIM radix sorts + a boundary scan;
//   f4  quality metrics of a [users x k] recommendation block against a ground-truth CSR, the per-user formulas of
//       replay/metrics/{ndcg,hitrate,precision,recall,map,mrr}.py::_get_metric_value_by_user.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

static inline int64_t a256(int64_t x) { return (x + 255) / 256 * 256; }

// =============================================================================================================
// f2  CSR builder
// =============================================================================================================
__global__ void iota_kernel(uint32_t* p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (uint32_t)i;
}
template <typename K, typename S>
__global__ void gather_key_kernel(const S* __restrict__ src, const uint32_t* __restrict__ perm, int64_t n, K* dst,
                                  K bias) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (K)src[perm[i]] + bias;
}
__global__ void csr_finish_kernel(const int32_t* __restrict__ user_idx, const int32_t* __restrict__ item_idx,
                                  const double* __restrict__ relevance, const uint32_t* __restrict__ perm, int64_t n,
                                  int64_t n_users, int64_t* __restrict__ offsets, int32_t* __restrict__ items,
                                  float* __restrict__ rewards) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  // offsets[u] = first sorted position whose user >= u: fill the gap between consecutive users
  const int64_t cur = (i < n) ? (int64_t)user_idx[perm[i]] : n_users;
  const int64_t prev = (i > 0) ? (int64_t)user_idx[perm[i - 1]] : -1;
  for (int64_t u = prev + 1; u <= cur; ++u) offsets[u] = i;
  if (i < n) {
    items[i] = item_idx[perm[i]];
    if (rewards) rewards[i] = (float)relevance[perm[i]];
  }
}

extern "C" int64_t cqlrec_build_csr_ws_bytes(int64_t n_rows) {
  return 2 * a256(n_rows * 4) + 2 * a256(n_rows * 8) + a256(4 * n_rows * 8) + (16ll << 20) + 256;
}

extern "C" int cqlrec_build_csr(const int32_t* user_idx, const int32_t* item_idx, const int64_t* timestamp,
                                const double* relevance, int64_t n_rows, int64_t n_users, void* ws, int64_t ws_bytes,
                                int64_t* offsets, int32_t* items, float* rewards, cqlrec_stream stream) {
  // timestamp == NULL: rows ordered by (user, item) -- the per-user ascending `seen` lists of cqlrec_score_topk;
  // relevance == NULL (then rewards may be NULL too): no reward column is produced
  CQL_REQUIRE(user_idx && item_idx && ws && offsets && items, "build_csr: NULL pointer");
  CQL_REQUIRE((relevance != nullptr) == (rewards != nullptr), "build_csr: relevance and rewards go together");
  CQL_REQUIRE(n_rows > 0 && n_rows < (1ll << 32) && n_users > 0, "build_csr: n_rows=%lld n_users=%lld",
              (long long)n_rows, (long long)n_users);
  CQL_REQUIRE(ws_bytes >= cqlrec_build_csr_ws_bytes(n_rows), "build_csr: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  char* p = (char*)ws;
  uint32_t* perm_a = (uint32_t*)p;  p += a256(n_rows * 4);
  uint32_t* perm_b = (uint32_t*)p;  p += a256(n_rows * 4);
  uint64_t* key_a = (uint64_t*)p;   p += a256(n_rows * 8);
  uint64_t* key_b = (uint64_t*)p;   p += a256(n_rows * 8);
  void* temp = p;
  size_t temp_cap = (size_t)(a256(4 * n_rows * 8) + (16ll << 20));
  const dim3 grid(cql_ceil_div(n_rows, 256)), block(256);
  hipLaunchKernelGGL(iota_kernel, grid, block, 0, s, perm_a, n_rows);
  // LSD over the composite key: item (least significant), then timestamp, then user; every pass is a STABLE sort
  auto pass = [&](int which) -> int {
    int bits = 64;
    if (which == 0) {
      hipLaunchKernelGGL((gather_key_kernel<uint64_t, int32_t>), grid, block, 0, s, item_idx, perm_a, n_rows, key_a,
                         (uint64_t)0);
      bits = 32;
    } else if (which == 1) {   // signed timestamps -> order-preserving unsigned
      hipLaunchKernelGGL((gather_key_kernel<uint64_t, int64_t>), grid, block, 0, s, timestamp, perm_a, n_rows, key_a,
                         (uint64_t)1 << 63);
    } else {
      hipLaunchKernelGGL((gather_key_kernel<uint64_t, int32_t>), grid, block, 0, s, user_idx, perm_a, n_rows, key_a,
                         (uint64_t)0);
      bits = 32;
    }
    size_t need = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, need, key_a, key_b, perm_a, perm_b, (size_t)n_rows, 0u,
                                             (unsigned)bits, s);
    if (e != hipSuccess || need > temp_cap) {
      cql_set_error("build_csr: radix sort needs %zu bytes of scratch (have %zu), err=%d", need, temp_cap, (int)e);
      return CQLREC_ERR_HIP;
    }
    e = rocprim::radix_sort_pairs(temp, need, key_a, key_b, perm_a, perm_b, (size_t)n_rows, 0u, (unsigned)bits, s);
    if (e != hipSuccess) {
      cql_set_error("build_csr: radix sort failed: %s", hipGetErrorString(e));
      return CQLREC_ERR_HIP;
    }
    uint32_t* t = perm_a;
    perm_a = perm_b;
    perm_b = t;
    return CQLREC_OK;
  };
  for (int w = 0; w < 3; ++w) {
    if (w == 1 && !timestamp) continue;
    const int rc = pass(w);
    if (rc != CQLREC_OK) return rc;
  }
  hipLaunchKernelGGL(csr_finish_kernel, dim3(cql_ceil_div(n_rows + 1, 256)), block, 0, s, user_idx, item_idx, relevance,
                     perm_a, n_rows, n_users, offsets, items, rewards);
  CQL_LAUNCH_CHECK("build_csr");
  return CQLREC_OK;
}

// =============================================================================================================
// f4  top-k quality metrics
// =============================================================================================================
#define EV_MAX_KS 8
struct EvKs {
  int32_t k[EV_MAX_KS];
  int32_t n;
};

__global__ __launch_bounds__(256) void eval_topk_kernel(const int32_t* __restrict__ rec_idx, int64_t n_users, int kmax,
                                                        const int32_t* __restrict__ rec_rows,
                                                        const int64_t* __restrict__ gt_off,
                                                        const int32_t* __restrict__ gt_items, EvKs ks,
                                                        double* __restrict__ per_user, double* __restrict__ block_sums) {
  __shared__ double red[256];
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double vals[CQLREC_EVAL_METRICS][EV_MAX_KS];
#pragma unroll
  for (int m = 0; m < CQLREC_EVAL_METRICS; ++m)
#pragma unroll
    for (int q = 0; q < EV_MAX_KS; ++q) vals[m][q] = 0.0;
  if (u < n_users) {
    const int64_t row = rec_rows ? (int64_t)rec_rows[u] : u;
    const int64_t g0 = gt_off[row];
    const int ngt = (int)(gt_off[row + 1] - g0);
    int npred = 0;
    while (npred < kmax && rec_idx[u * kmax + npred] >= 0) ++npred;
    int hits = 0, first = -1;
    double dcg = 0.0, ap = 0.0, idcg = 0.0;
    int q = 0;
    for (int j = 0; j < kmax && q < ks.n; ++j) {
      const double w = 1.0 / log2((double)(j + 2));
      if (j < ngt) idcg += w;
      if (j < npred && ngt > 0) {
        const int32_t it = rec_idx[u * kmax + j];
        int lo = 0, hi = ngt;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (gt_items[g0 + mid] < it) lo = mid + 1; else hi = mid;
        }
        if (lo < ngt && gt_items[g0 + lo] == it) {
          ++hits;
          dcg += w;
          ap += (double)hits / (double)(j + 1);
          if (first < 0) first = j;
        }
      }
      while (q < ks.n && ks.k[q] == j + 1) {   // metrics at k = j+1 (ks ascending)
        const int k = j + 1;
        const bool ok = npred > 0 && ngt > 0;
        vals[0][q] = ok ? dcg / idcg : 0.0;                              // ndcg.py:50-59
        vals[1][q] = hits > 0 ? 1.0 : 0.0;                               // hitrate.py:22-27
        vals[2][q] = npred > 0 ? (double)hits / (double)k : 0.0;         // precision.py
        vals[3][q] = ngt > 0 ? (double)hits / (double)ngt : 0.0;         // recall.py
        vals[4][q] = ok ? ap / (double)k : 0.0;                          // map.py
        vals[5][q] = first >= 0 ? 1.0 / (double)(first + 1) : 0.0;       // mrr.py
        ++q;
      }
    }
    if (per_user) {
      for (int m = 0; m < CQLREC_EVAL_METRICS; ++m)
        for (int qq = 0; qq < ks.n; ++qq) per_user[(u * CQLREC_EVAL_METRICS + m) * ks.n + qq] = vals[m][qq];
    }
  }
  // deterministic block sums
  for (int m = 0; m < CQLREC_EVAL_METRICS; ++m) {
    for (int qq = 0; qq < ks.n; ++qq) {
      red[threadIdx.x] = vals[m][qq];
      __syncthreads();
      for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
      }
      if (threadIdx.x == 0) block_sums[((int64_t)blockIdx.x * CQLREC_EVAL_METRICS + m) * ks.n + qq] = red[0];
      __syncthreads();
    }
  }
}

__global__ void eval_final_kernel(const double* __restrict__ block_sums, int nblocks, int nvals, double* __restrict__ sums) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nvals) return;
  double s = 0.0;
  for (int b = 0; b < nblocks; ++b) s += block_sums[(int64_t)b * nvals + v];
  sums[v] = s;
}

extern "C" int64_t cqlrec_eval_topk_ws_bytes(int64_t n_users, int32_t n_ks) {
  return a256((int64_t)cql_ceil_div(n_users, 256) * CQLREC_EVAL_METRICS * n_ks * 8) + 256;
}

extern "C" int cqlrec_eval_topk(const int32_t* rec_idx, int64_t n_users, int32_t kmax, const int32_t* rec_rows,
                                const int64_t* gt_off, const int32_t* gt_items, const int32_t* ks, int32_t n_ks,
                                void* ws, int64_t ws_bytes, double* per_user, double* sums, cqlrec_stream stream) {
  CQL_REQUIRE(rec_idx && gt_off && gt_items && ks && ws && sums, "eval_topk: NULL pointer");
  CQL_REQUIRE(n_users > 0 && kmax > 0, "eval_topk: n_users=%lld kmax=%d", (long long)n_users, kmax);
  CQL_REQUIRE(n_ks > 0 && n_ks <= EV_MAX_KS, "eval_topk: n_ks=%d out of range (1..%d)", n_ks, EV_MAX_KS);
  CQL_REQUIRE(ws_bytes >= cqlrec_eval_topk_ws_bytes(n_users, n_ks), "eval_topk: workspace too small");
  EvKs e;
  e.n = n_ks;
  for (int i = 0; i < n_ks; ++i) {
    CQL_REQUIRE(ks[i] > 0 && ks[i] <= kmax && (i == 0 || ks[i] > ks[i - 1]),
                "eval_topk: ks must be ascending and within 1..kmax");
    e.k[i] = ks[i];
  }
  for (int i = n_ks; i < EV_MAX_KS; ++i) e.k[i] = 0;
  hipStream_t s = (hipStream_t)stream;
  const int nblocks = cql_ceil_div(n_users, 256);
  hipLaunchKernelGGL(eval_topk_kernel, dim3(nblocks), dim3(256), 0, s, rec_idx, n_users, kmax, rec_rows, gt_off, gt_items,
                     e, per_user, (double*)ws);
  const int nvals = CQLREC_EVAL_METRICS * n_ks;
  hipLaunchKernelGGL(eval_final_kernel, dim3(1), dim3(64), 0, s, (const double*)ws, nblocks, nvals, sums);
  CQL_LAUNCH_CHECK("eval_topk");
  return CQLREC_OK;
}
