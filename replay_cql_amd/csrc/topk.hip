// All-users top-K scoring (S7) without materialising the users x items score matrix.
//
//   pass 1  qstream_kernel<TILEMAX>: the MFMA Q-head; per (user, group of G=32*tg items) only the group maximum is
//           written:  tilemax[group][user]  (n_cand/G floats per user instead of n_cand).
//   pass 2  topk_select_kernel (one wave per user):
//           a. radix-select the K' = k + n_seen(user) best groups by (max desc, group asc).  Every admissible item that
//              belongs to the final top-k lies in one of them: each selected group holds an element >= the item, at
//              most n_seen of those elements are excluded ones, and ties resolve towards the lower group/item id.
//           b. re-score exactly those groups with the same MFMA chain (bit-identical to pass 1), drop seen /
//              out-of-range items, keep candidates as 64-bit keys (order-preserving score bits << 32 | ~item id);
//           c. radix-select the k largest keys, rank them, write (item id, score).
//   Ordering is exactly (score desc, item id asc) -- the tie rule of SURVEY.md F7 / 8.0 S7.
#include "qhead_internal.h"

#define TK_CB 2048          // candidate buffer entries (LDS)
#define TK_MAX_K 1024
#define TK_MAX_GROUPS 4096

__device__ __forceinline__ uint64_t make_key(float score, uint32_t id) {
  return ((uint64_t)f32_order_key(score) << 32) | (uint64_t)(~id);
}

// k-th largest (1-based) of n distinct 64-bit keys in LDS `buf`; whole wave participates; hist = 256 LDS words.
__device__ uint64_t radix_kth(const uint64_t* buf, int n, int kth, uint32_t* hist, int lane) {
  uint64_t prefix = 0;
  int need = kth;
  for (int shift = 56; shift >= 0; shift -= 8) {
    for (int i = lane; i < 256; i += 64) hist[i] = 0;
    __syncthreads();
    for (int i = lane; i < n; i += 64) {
      const uint64_t key = buf[i];
      const bool match = (shift == 56) || ((key >> (shift + 8)) == (prefix >> (shift + 8)));
      if (match) atomicAdd(&hist[(key >> shift) & 255], 1u);
    }
    __syncthreads();
    uint32_t bins[4];
    uint32_t local = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      bins[b] = hist[lane * 4 + b];
      local += bins[b];
    }
    uint32_t suf = local;  // inclusive suffix sum over lanes >= lane
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_down(suf, off);
      if (lane + off < 64) suf += t;
    }
    const uint32_t above = suf - local;
    const bool mine = (above < (uint32_t)need) && ((uint32_t)need <= suf);
    int digit = 0, need_new = need;
    if (mine) {
      uint32_t c = above;
#pragma unroll
      for (int b = 3; b >= 0; --b) {
        if (c + bins[b] >= (uint32_t)need) {
          digit = lane * 4 + b;
          need_new = need - (int)c;
          break;
        }
        c += bins[b];
      }
    }
    const unsigned long long m = __ballot(mine);
    const int src = __ffsll((long long)m) - 1;
    digit = __shfl(digit, src);
    need = __shfl(need_new, src);
    prefix |= (uint64_t)digit << shift;
    __syncthreads();
  }
  return prefix;
}

// keep the k largest keys of buf[0..n) at the front (unordered); returns the new count
__device__ int select_topk_inplace(uint64_t* buf, int n, int k, uint32_t* hist, int lane) {
  if (n <= k) return n;
  const uint64_t thr = radix_kth(buf, n, k, hist, lane);
  int cnt = 0;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    const uint64_t key = (i < n) ? buf[i] : 0;
    const bool keep = (i < n) && (key >= thr);
    const unsigned long long m = __ballot(keep);
    const int pos = cnt + __popcll(m & ((1ull << lane) - 1));
    __syncthreads();
    if (keep) buf[pos] = key;
    cnt += __popcll(m);
    __syncthreads();
  }
  return cnt;
}

template <int D>
__global__ __launch_bounds__(64) void topk_select_kernel(const uint16_t* __restrict__ H_b, int64_t n_users,
                                                         const uint16_t* __restrict__ E_b, const float* __restrict__ b,
                                                         int64_t n_cand, const int32_t* __restrict__ item_ids,
                                                         const int64_t* __restrict__ seen_off,
                                                         const int32_t* __restrict__ seen_items,
                                                         const int32_t* __restrict__ seen_rows,
                                                         const float* __restrict__ tilemax, int ngroups, int tg, int k,
                                                         int32_t* __restrict__ out_idx, float* __restrict__ out_val,
                                                         int32_t* __restrict__ out_cnt) {
  constexpr int KS = D / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem);                 // 256 words
  float* scores = reinterpret_cast<float*>(smem + 1024);              // 32 floats
  uint64_t* buf = reinterpret_cast<uint64_t*>(smem + 1024 + 128);     // max(ngroups, K' ints + TK_CB keys)

  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int64_t u = blockIdx.x;
  const int64_t srow = seen_rows ? (int64_t)seen_rows[u] : u;
  const int64_t so = seen_off ? seen_off[srow] : 0;
  const int ns = seen_off ? (int)(seen_off[srow + 1] - so) : 0;
  int kprime = k + ns;
  if (kprime > ngroups) kprime = ngroups;

  // ---- a. select the K' best groups ------------------------------------------------------------------------
  for (int g = lane; g < ngroups; g += 64) buf[g] = make_key(tilemax[(int64_t)g * n_users + u], (uint32_t)g);
  __syncthreads();
  uint64_t thr = 0;
  if (kprime < ngroups) thr = radix_kth(buf, ngroups, kprime, hist, lane);
  int32_t* sel = reinterpret_cast<int32_t*>(buf);
  int nsel = 0;
  for (int base = 0; base < ngroups; base += 64) {
    const int g = base + lane;
    const bool keep = (g < ngroups) && (buf[g] >= thr);
    const unsigned long long m = __ballot(keep);
    const int pos = nsel + __popcll(m & ((1ull << lane) - 1));
    __syncthreads();
    if (keep) sel[pos] = g;   // pos <= g: lands in bytes of keys already consumed
    nsel += __popcll(m);
    __syncthreads();
  }
  uint64_t* cand = reinterpret_cast<uint64_t*>(reinterpret_cast<unsigned char*>(buf) + ((nsel * 4 + 15) / 16) * 16);
  int ncand = 0;

  // ---- b. exact re-scoring of the selected groups -----------------------------------------------------------
  bf16x8 hf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) hf[s] = *reinterpret_cast<const bf16x8*>(H_b + u * D + 16 * s + 8 * h);

  for (int si = 0; si < nsel; ++si) {
    const int g = sel[si];
    for (int t = 0; t < tg; ++t) {
      const int64_t item0 = ((int64_t)g * tg + t) * 32;
      if (item0 >= n_cand) break;
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int64_t c = item0 + mfma_row(i, h);
        acc[i] = (c < n_cand) ? b[c] : NEG_INF_F;
      }
      int64_t arow = item0 + r;
      if (arow >= n_cand) arow = n_cand - 1;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(E_b + arow * D + 16 * s + 8 * h);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, hf[s], acc, 0, 0, 0);
      }
      if (r == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) scores[mfma_row(i, h)] = acc[i];
      }
      __syncthreads();
      bool valid = false;
      uint64_t key = 0;
      if (lane < 32) {
        const int64_t c = item0 + lane;
        if (c < n_cand) {
          const float sc = scores[lane];
          const int32_t gid = item_ids ? item_ids[c] : (int32_t)c;
          valid = true;
          if (ns > 0) {  // binary search in the user's ascending seen list
            int lo = 0, hi = ns;
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              const int32_t v = seen_items[so + mid];
              if (v < gid) lo = mid + 1; else hi = mid;
            }
            if (lo < ns && seen_items[so + lo] == gid) valid = false;
          }
          key = make_key(sc, (uint32_t)gid);
        }
      }
      const unsigned long long m = __ballot(valid);
      const int pos = ncand + __popcll(m & ((1ull << lane) - 1));
      if (valid) cand[pos] = key;
      ncand += __popcll(m);
      __syncthreads();
      if (ncand + 32 > TK_CB) ncand = select_topk_inplace(cand, ncand, k, hist, lane);
    }
  }

  // ---- c. final selection + ranking ---------------------------------------------------------------------------
  ncand = select_topk_inplace(cand, ncand, k, hist, lane);
  for (int i = lane; i < ncand; i += 64) {
    const uint64_t key = cand[i];
    int rank = 0;
    for (int j = 0; j < ncand; ++j) rank += (cand[j] > key) ? 1 : 0;
    out_idx[u * k + rank] = (int32_t)(~(uint32_t)(key & 0xFFFFFFFFull));
    out_val[u * k + rank] = f32_from_order_key((uint32_t)(key >> 32));
  }
  for (int i = ncand + lane; i < k; i += 64) {
    out_idx[u * k + i] = -1;
    out_val[u * k + i] = NEG_INF_F;
  }
  if (lane == 0) out_cnt[u] = ncand;
}

// =============================================================================================================
static inline int64_t align256(int64_t x) { return (x + 255) / 256 * 256; }

static int tk_tile_groups(int64_t n_cand, int* tg_out) {
  const int64_t tiles = (n_cand + 31) / 32;
  int tg = 1;
  while ((tiles + tg - 1) / tg > TK_MAX_GROUPS) tg *= 2;
  *tg_out = tg;
  return (int)((tiles + tg - 1) / tg);
}

extern "C" int64_t cqlrec_topk_ws_bytes(int64_t n_users, int64_t n_cand, int32_t d, int32_t k) {
  (void)d;
  (void)k;
  int tg;
  const int ngroups = tk_tile_groups(n_cand, &tg);
  return align256((int64_t)ngroups * n_users * 4) + 256;
}

extern "C" int cqlrec_score_topk(const uint16_t* H_b, int64_t n_users, const uint16_t* E_b, const float* b,
                                 int64_t n_cand, int32_t d, const int32_t* item_ids, const int64_t* seen_off,
                                 const int32_t* seen_items, const int32_t* seen_rows, int32_t k, void* ws,
                                 int64_t ws_bytes, int32_t* out_idx, float* out_val, int32_t* out_cnt,
                                 cqlrec_stream stream) {
  CQL_REQUIRE(H_b && E_b && b && ws && out_idx && out_val && out_cnt, "score_topk: NULL pointer");
  CQL_REQUIRE(d == 64 || d == 128 || d == 256, "score_topk: d=%d unsupported", d);
  CQL_REQUIRE(n_users > 0 && n_cand > 0, "score_topk: n_users=%lld n_cand=%lld", (long long)n_users, (long long)n_cand);
  CQL_REQUIRE(k > 0 && k <= TK_MAX_K, "score_topk: k=%d out of range (1..%d)", k, TK_MAX_K);
  CQL_REQUIRE(seen_off == nullptr || seen_items != nullptr, "score_topk: seen_items is NULL");
  CQL_REQUIRE(ws_bytes >= cqlrec_topk_ws_bytes(n_users, n_cand, d, k), "score_topk: workspace too small");
  int tg;
  const int ngroups = tk_tile_groups(n_cand, &tg);
  hipStream_t s = (hipStream_t)stream;
  // pass 1
  const int unit = (32 * tg > QS_TI) ? 32 * tg : QS_TI;
  const QSplit sp = qs_choose_split(n_cand, n_users, QS_SPW_FWD, unit);
  QArgs a = {};
  a.res = H_b;
  a.n_res = n_users;
  a.str = E_b;
  a.n_str = n_cand;
  a.str_scalar = b;
  a.nsplit = sp.nsplit;
  a.split_rows = sp.split_rows;
  a.tilemax = (float*)ws;
  a.tg = tg;
  qs_launch(QM_TILEMAX, a, d, sp.rblks, s);
  // pass 2
  const int64_t buf_bytes_a = (int64_t)ngroups * 8;
  const int64_t buf_bytes_b = (int64_t)(((int64_t)ngroups * 4 + 15) / 16 * 16) + (int64_t)TK_CB * 8;
  const size_t smem = (size_t)(1024 + 128 + (buf_bytes_a > buf_bytes_b ? buf_bytes_a : buf_bytes_b));
  dim3 grid((unsigned)n_users), block(64);
  CqlProfScope prof(CQLREC_PH_TOPK_SELECT, s);
#define TK_LAUNCH(DD)                                                                                                 \
  hipLaunchKernelGGL(topk_select_kernel<DD>, grid, block, smem, s, H_b, n_users, E_b, b, n_cand, item_ids, seen_off, \
                     seen_items, seen_rows, (const float*)ws, ngroups, tg, k, out_idx, out_val, out_cnt)
  if (d == 64) TK_LAUNCH(64); else if (d == 128) TK_LAUNCH(128); else TK_LAUNCH(256);
#undef TK_LAUNCH
  CQL_LAUNCH_CHECK("score_topk");
  return CQLREC_OK;
}
