// All-users top-K scoring (S7) without materialising the users x items score matrix.
//
//   pass 1  qstream_kernel<TILEMAX>: the MFMA Q-head; per (user, group of G = 32*tg items) only the group maximum
//           is written, tilemax[group][user] (n_cand/G floats per user instead of n_cand), then transposed to
//           [user][group] so that the per-user pass reads whole lines.
//   pass 2  topk_select_kernel, one wave per user, exact "threshold algorithm":
//           round 1: radix-select the k best groups by (max desc, group asc) from keys held in REGISTERS, re-score
//                    those groups with the same MFMA chain as pass 1 (bit-identical scores), drop seen / out-of-range
//                    items, keep candidates as 64-bit keys (order-preserving score bits << 32 | ~item id);
//           round n: tau = k-th best candidate so far.  Only a group whose upper bound (its maximum, at its first item
//                    id) still beats tau can change the answer; re-score exactly those, update tau, repeat.  Stops
//                    when no unprocessed group can beat tau -- usually after k + (a few) groups of 32 items.
//           finally rank the k survivors and write (item id, score).
//   Ordering is exactly (score desc, item id asc) -- the tie rule of SURVEY.md F7 / 8.0 S7.
#include <stdlib.h>
#include "qhead_internal.h"

#define TK_CB_SMALL 1024    // candidate buffer entries (LDS, 8 KiB): k <= 512
#define TK_CB_LARGE 4096    // 32 KiB: k <= 2048 (rare; lower occupancy)
#define TK_MAX_K 2048
#define TK_MAX_GROUPS 4096
#define TK_SEEN_LDS 512      // seen-list entries kept in LDS per user

__device__ __forceinline__ uint64_t make_key(float score, uint32_t id) {
  return ((uint64_t)f32_order_key(score) << 32) | (uint64_t)(~id);
}

// digit selection shared by the two radix selects: with the histogram of the current digit in hist[], find the digit
// that holds the `need`-th largest element; returns (digit, rank inside that digit's bin) wave-uniformly.
__device__ __forceinline__ void radix_pick(const uint32_t* hist, int lane, int& need, int& digit) {
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
  int dg = 0, need_new = need;
  if (mine) {
    uint32_t c = above;
#pragma unroll
    for (int b = 3; b >= 0; --b) {
      if (c + bins[b] >= (uint32_t)need) {
        dg = lane * 4 + b;
        need_new = need - (int)c;
        break;
      }
      c += bins[b];
    }
  }
  const unsigned long long m = __ballot(mine);
  const int src = __ffsll((long long)m) - 1;
  digit = __shfl(dg, src);
  need = __shfl(need_new, src);
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
    int digit;
    radix_pick(hist, lane, need, digit);
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

// [groups][users] -> [users][gstride]  (32x32 tiles through LDS; both sides coalesced)
__global__ __launch_bounds__(256) void tilemax_transpose_kernel(const float* __restrict__ src, int ngroups,
                                                                int64_t n_users, float* __restrict__ dst, int gstride) {
  __shared__ float t[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int64_t u0 = (int64_t)blockIdx.x * 32;
  const int g0 = blockIdx.y * 32;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g = g0 + ty + 8 * j;
    const int64_t u = u0 + tx;
    t[ty + 8 * j][tx] = (g < ngroups && u < n_users) ? src[(int64_t)g * n_users + u] : NEG_INF_F;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t u = u0 + ty + 8 * j;
    const int g = g0 + tx;
    if (u < n_users && g < gstride) dst[u * gstride + g] = t[tx][ty + 8 * j];
  }
}

template <int D, int KPL, int TK_CB>
__global__ __launch_bounds__(64) void topk_select_kernel(const uint16_t* __restrict__ H_b, int64_t n_users,
                                                         const uint16_t* __restrict__ E_b, const float* __restrict__ b,
                                                         int64_t n_cand, const int32_t* __restrict__ item_ids,
                                                         const int64_t* __restrict__ seen_off,
                                                         const int32_t* __restrict__ seen_items,
                                                         const int32_t* __restrict__ seen_rows,
                                                         const float* __restrict__ tm_t, int gstride, int ngroups,
                                                         int tg, int k, int32_t* __restrict__ out_idx,
                                                         float* __restrict__ out_val, int32_t* __restrict__ out_cnt) {
  constexpr int KS = D / 16;
  __shared__ uint32_t hist[256];
  __shared__ float scores[32];
  __shared__ uint64_t cand[TK_CB];
  __shared__ int32_t seen_lds[TK_SEEN_LDS];

  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int64_t u = blockIdx.x;
  const int64_t srow = seen_rows ? (int64_t)seen_rows[u] : u;
  const int64_t so = seen_off ? seen_off[srow] : 0;
  const int ns = seen_off ? (int)(seen_off[srow + 1] - so) : 0;
  const unsigned long long lt_mask = (1ull << lane) - 1;

  // ---- group maxima of this user -> order-preserving keys in registers (group g = slot*64 + lane) -----------
  uint32_t key[KPL];
#pragma unroll
  for (int sl = 0; sl < KPL; ++sl) {
    const int g = sl * 64 + lane;
    key[sl] = (g < ngroups) ? f32_order_key(tm_t[u * gstride + g]) : 0u;
  }

  // digits on which every key agrees need no histogram (scores of one user usually share the top byte; a shared
  // digit would also serialise all the LDS atomics on one word)
  uint32_t k_and = 0xFFFFFFFFu, k_or = 0u;
#pragma unroll
  for (int sl = 0; sl < KPL; ++sl) {
    if (sl * 64 + lane < ngroups) {
      k_and &= key[sl];
      k_or |= key[sl];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    k_and &= __shfl_xor(k_and, off);
    k_or |= __shfl_xor(k_or, off);
  }
  const uint32_t k_diff = k_and ^ k_or;
  unsigned long long done = 0;   // bit sl: this lane's slot sl has been re-scored
  int remaining = ngroups;

  // radix select over the NOT YET PROCESSED keys: T = want-th largest value among them, need_eq = how many of the
  // == T keys (in group order) belong to the `want` best
  auto select_round = [&](int want, uint32_t& T, int& need_eq) {
    T = 0;
    int need = want;
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (((k_diff >> shift) & 255u) == 0u) {
        T |= k_and & (255u << shift);
        continue;
      }
      for (int i = lane; i < 256; i += 64) hist[i] = 0;
      __syncthreads();
#pragma unroll
      for (int sl = 0; sl < KPL; ++sl) {
        const int g = sl * 64 + lane;
        const bool match = (g < ngroups) && !((done >> sl) & 1ull) &&
                           ((shift == 24) || ((key[sl] >> (shift + 8)) == (T >> (shift + 8))));
        if (match) atomicAdd(&hist[(key[sl] >> shift) & 255], 1u);
      }
      __syncthreads();
      int digit;
      radix_pick(hist, lane, need, digit);
      T |= (uint32_t)digit << shift;
      __syncthreads();
    }
    need_eq = need;
  };

  // the user's ascending seen list in LDS (binary-searched once per candidate); longer lists stay in global memory
  const bool seen_in_lds = ns <= TK_SEEN_LDS;
  if (seen_in_lds) {
    for (int i = lane; i < ns; i += 64) seen_lds[i] = seen_items[so + i];
  }
  __syncthreads();

  bf16x8 hf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) hf[s] = *reinterpret_cast<const bf16x8*>(H_b + u * D + 16 * s + 8 * h);

  int ncand = 0;
  // exact re-scoring of one group (wave-uniform g): tg tiles of 32 candidates -> admissible ones appended to cand[]
  auto rescore = [&](int g) {
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
      uint64_t ck = 0;
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
              const int32_t v = seen_in_lds ? seen_lds[mid] : seen_items[so + mid];
              if (v < gid) lo = mid + 1; else hi = mid;
            }
            if (lo < ns && (seen_in_lds ? seen_lds[lo] : seen_items[so + lo]) == gid) valid = false;
          }
          ck = make_key(sc, (uint32_t)gid);
        }
      }
      const unsigned long long m = __ballot(valid);
      const int pos = ncand + __popcll(m & lt_mask);
      if (valid) cand[pos] = ck;
      ncand += __popcll(m);
      __syncthreads();
      if (ncand + 32 > TK_CB) ncand = select_topk_inplace(cand, ncand, k, hist, lane);
    }
  };

  // ---- rounds: the next k best unprocessed groups by (max desc, group asc), as long as one of them can still beat
  //      the k-th best admissible candidate found so far (groups are visited in descending order of their bound) ----
  uint64_t tau = 0;   // 0: fewer than k admissible candidates yet -> every group qualifies
  while (remaining > 0) {
    const uint32_t tau_hi = (uint32_t)(tau >> 32);
    if (tau != 0) {   // cheap exit: the best unprocessed bound is already below the k-th best candidate
      uint32_t rem_max = 0;
#pragma unroll
      for (int sl = 0; sl < KPL; ++sl) {
        const bool in = (sl * 64 + lane < ngroups) && !((done >> sl) & 1ull);
        rem_max = (in && key[sl] > rem_max) ? key[sl] : rem_max;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(rem_max, off);
        rem_max = o > rem_max ? o : rem_max;
      }
      if (rem_max < tau_hi) break;
    }
    const int want = k < remaining ? k : remaining;
    uint32_t T;
    int need_eq;
    select_round(want, T, need_eq);
    int eq_base = 0;
    bool any = false;
#pragma unroll
    for (int sl = 0; sl < KPL; ++sl) {
      const int g = sl * 64 + lane;
      const bool in = (g < ngroups) && !((done >> sl) & 1ull);
      const bool eq = in && key[sl] == T;
      const unsigned long long em = __ballot(eq);
      bool pick = in && (key[sl] > T || (eq && eq_base + __popcll(em & lt_mask) < need_eq));
      eq_base += __popcll(em);
      if (pick) {   // does the group's upper bound (its maximum, at its smallest item id) beat tau?
        pick = key[sl] >= tau_hi;
        if (pick && key[sl] == tau_hi && tau != 0) {
          const int64_t c0 = (int64_t)g * tg * 32;
          const uint32_t gid0 = (c0 < n_cand) ? (uint32_t)(item_ids ? item_ids[c0] : (int32_t)c0) : 0xFFFFFFFFu;
          pick = (((uint64_t)tau_hi << 32) | (uint64_t)(~gid0)) > tau;
        }
      }
      unsigned long long pm = __ballot(pick);
      if (pick) done |= 1ull << sl;
      any |= (pm != 0);
      remaining -= __popcll(pm);
      while (pm) {
        const int j = __ffsll((long long)pm) - 1;
        pm &= pm - 1;
        rescore(sl * 64 + j);
      }
    }
    if (!any) break;   // the best remaining groups cannot change the answer, nor can any worse one
    ncand = select_topk_inplace(cand, ncand, k, hist, lane);
    if (ncand >= k) {
      tau = ~0ull;
      for (int i = lane; i < ncand; i += 64) tau = cand[i] < tau ? cand[i] : tau;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const uint64_t o = __shfl_xor(tau, off);
        tau = o < tau ? o : tau;
      }
    }
  }
  ncand = select_topk_inplace(cand, ncand, k, hist, lane);

  // ---- rank the survivors -------------------------------------------------------------------------------------------
  for (int i = lane; i < ncand; i += 64) {
    const uint64_t ck = cand[i];
    int rank = 0;
    for (int j = 0; j < ncand; ++j) rank += (cand[j] > ck) ? 1 : 0;
    out_idx[u * k + rank] = (int32_t)(~(uint32_t)(ck & 0xFFFFFFFFull));
    out_val[u * k + rank] = f32_from_order_key((uint32_t)(ck >> 32));
  }
  for (int i = ncand + lane; i < k; i += 64) {
    out_idx[u * k + i] = -1;
    out_val[u * k + i] = NEG_INF_F;
  }
  if (lane == 0) out_cnt[u] = ncand;
}

// =============================================================================================================
// k <= 16: the common case, with far fewer instructions per user than the radix-select kernel above.
//   * every lane keeps the best TWO of its unprocessed group keys; the next-best group of the user is a wave-wide
//     max over the lanes' bests (when a lane has given away both, all lanes refill -- rare);
//   * groups are visited strictly in (max desc, group asc) order, one at a time, until the bound of the next one
//     cannot beat tau -- the threshold algorithm in its plain form;
//   * "the k best of the candidates" is k rounds of wave-wide max over register-held keys, which also leaves them
//     sorted, so the final ranking is free.
// Same exactness argument and identical results as topk_select_kernel (tests run both).
// =============================================================================================================
#define TKS_MAX_K 16
#define TKS_CB 512           // candidate capacity in LDS = 8 register slots per lane

__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint64_t o = __shfl_xor(v, off);
    v = o > v ? o : v;
  }
  return v;
}

template <int D, int KPL>
__global__ __launch_bounds__(64) void topk_select_small_kernel(const uint16_t* __restrict__ H_b, int64_t n_users,
                                                               const uint16_t* __restrict__ E_b,
                                                               const float* __restrict__ b, int64_t n_cand,
                                                               const int32_t* __restrict__ item_ids,
                                                               const int64_t* __restrict__ seen_off,
                                                               const int32_t* __restrict__ seen_items,
                                                               const int32_t* __restrict__ seen_rows,
                                                               const float* __restrict__ tm_t, int gstride, int ngroups,
                                                               int tg, int k, int32_t* __restrict__ out_idx,
                                                               float* __restrict__ out_val, int32_t* __restrict__ out_cnt) {
  constexpr int KS = D / 16;
  __shared__ __attribute__((aligned(16))) float scores[32];
  __shared__ uint64_t cand[TKS_CB];
  __shared__ uint64_t best[TKS_MAX_K];
  __shared__ int32_t seen_lds[TK_SEEN_LDS];

  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int64_t u = blockIdx.x;
  const int64_t srow = seen_rows ? (int64_t)seen_rows[u] : u;
  const int64_t so = seen_off ? seen_off[srow] : 0;
  const int ns = seen_off ? (int)(seen_off[srow + 1] - so) : 0;
  const unsigned long long lt_mask = (1ull << lane) - 1;

  const bool seen_in_lds = ns <= TK_SEEN_LDS;
  if (seen_in_lds) {
    for (int i = lane; i < ns; i += 64) seen_lds[i] = seen_items[so + i];
  }
  bf16x8 hf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) hf[s] = *reinterpret_cast<const bf16x8*>(H_b + u * D + 16 * s + 8 * h);
  __syncthreads();

  // ---- per-lane best three unprocessed groups (key, slot); key 0 = none.  The group maxima themselves stay in
  // memory (one coalesced sweep of the user's row here; again only when a lane has handed out all three and still owns
  // more): keeping the <= 64 keys of a lane in registers cost 64 VGPRs and half of the kernel's occupancy.
  const int nvalid = (ngroups > lane) ? (ngroups - lane + 63) / 64 : 0;   // slots of this lane that hold a group
  unsigned long long done = 0;
  uint32_t b1k = 0, b2k = 0, b3k = 0;
  int b1s = 0, b2s = 0, b3s = 0;
  const float* tm_row = tm_t + u * gstride;
  auto refill = [&]() {
    b1k = 0; b2k = 0; b3k = 0; b1s = 0; b2s = 0; b3s = 0;
#pragma unroll 8
    for (int sl = 0; sl < KPL; ++sl) {
      const int g = sl * 64 + lane;
      uint32_t kk = 0u;                                   // real keys are > 0
      if (g < ngroups && !((done >> sl) & 1ull)) kk = f32_order_key(tm_row[g]);
      const bool gt1 = kk > b1k, gt2 = kk > b2k, gt3 = kk > b3k;   // ascending slots + strict '>' keep the earliest on ties
      b3k = gt2 ? b2k : (gt3 ? kk : b3k);
      b3s = gt2 ? b2s : (gt3 ? sl : b3s);
      b2k = gt1 ? b1k : (gt2 ? kk : b2k);
      b2s = gt1 ? b1s : (gt2 ? sl : b2s);
      b1k = gt1 ? kk : b1k;
      b1s = gt1 ? sl : b1s;
    }
  };
  refill();

  int ncand = 0;
  uint64_t tau = 0;        // k-th best candidate key (valid once have_k)
  bool have_k = false;

  // the k best of cand[0..ncand): k rounds of wave max over register-held keys -> best[] (descending), compacted back
  auto take_top_k = [&]() {
    uint64_t ck[TKS_CB / 64];
#pragma unroll
    for (int q = 0; q < TKS_CB / 64; ++q) ck[q] = (q * 64 + lane < ncand) ? cand[q * 64 + lane] : 0ull;
    const int kk = k < ncand ? k : ncand;
    for (int j = 0; j < kk; ++j) {
      uint64_t m = 0;
#pragma unroll
      for (int q = 0; q < TKS_CB / 64; ++q) m = ck[q] > m ? ck[q] : m;
      const uint64_t w = wave_max_u64(m);
#pragma unroll
      for (int q = 0; q < TKS_CB / 64; ++q) ck[q] = (ck[q] == w) ? 0ull : ck[q];   // keys are distinct
      if (lane == 0) best[j] = w;
    }
    __syncthreads();
    if (lane < kk) cand[lane] = best[lane];
    ncand = kk;
    have_k = (kk == k);
    tau = have_k ? best[k - 1] : 0ull;
    __syncthreads();
  };

  auto rescore = [&](int g) {
    for (int t = 0; t < tg; ++t) {
      const int64_t item0 = ((int64_t)g * tg + t) * 32;
      if (item0 >= n_cand) break;
      if (ncand + 32 > TKS_CB) take_top_k();
      int64_t arow = item0 + r;
      if (arow >= n_cand) arow = n_cand - 1;
      bf16x8 af[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) af[s] = *reinterpret_cast<const bf16x8*>(E_b + arow * D + 16 * s + 8 * h);
      if (h == 0) scores[r] = (item0 + r < n_cand) ? b[item0 + r] : NEG_INF_F;
      __syncthreads();
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 t4 = *reinterpret_cast<const float4*>(&scores[8 * q + 4 * h]);
        acc[4 * q + 0] = t4.x;
        acc[4 * q + 1] = t4.y;
        acc[4 * q + 2] = t4.z;
        acc[4 * q + 3] = t4.w;
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], hf[s], acc, 0, 0, 0);
      __syncthreads();
      if (r == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) scores[mfma_row(i, h)] = acc[i];
      }
      __syncthreads();
      bool valid = false;
      uint64_t ck = 0;
      if (lane < 32) {
        const int64_t c = item0 + lane;
        if (c < n_cand) {
          const int32_t gid = item_ids ? item_ids[c] : (int32_t)c;
          ck = make_key(scores[lane], (uint32_t)gid);
          valid = !have_k || ck > tau;            // below the current k-th best: cannot enter the answer
          if (valid && ns > 0) {                  // binary search in the user's ascending seen list
            int lo = 0, hi = ns;
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              const int32_t v = seen_in_lds ? seen_lds[mid] : seen_items[so + mid];
              if (v < gid) lo = mid + 1; else hi = mid;
            }
            if (lo < ns && (seen_in_lds ? seen_lds[lo] : seen_items[so + lo]) == gid) valid = false;
          }
        }
      }
      const unsigned long long m = __ballot(valid);
      if (valid) cand[ncand + __popcll(m & lt_mask)] = ck;
      ncand += __popcll(m);
      __syncthreads();
    }
  };

  // ---- groups in (max desc, group asc) order until the next bound cannot beat tau ----------------------------------
  int visited = 0;
  for (;;) {
    const uint64_t c1 = b1k ? (((uint64_t)b1k << 32) | (uint64_t)(~(uint32_t)(b1s * 64 + lane))) : 0ull;
    const uint64_t cw = wave_max_u64(c1);
    if (cw == 0) break;                                  // every group processed
    const int g = (int)(~(uint32_t)(cw & 0xFFFFFFFFull));
    if (have_k) {
      const uint32_t gk = (uint32_t)(cw >> 32), tau_hi = (uint32_t)(tau >> 32);
      bool beats = gk > tau_hi;
      if (gk == tau_hi) {                                // tie on the score: the group's smallest item id decides
        const int64_t c0 = (int64_t)g * tg * 32;
        const uint32_t gid0 = (c0 < n_cand) ? (uint32_t)(item_ids ? item_ids[c0] : (int32_t)c0) : 0xFFFFFFFFu;
        beats = (((uint64_t)gk << 32) | (uint64_t)(~gid0)) > tau;
      }
      if (!beats) break;
    }
    if (c1 == cw) {                                      // owner lane: pop
      done |= 1ull << b1s;
      b1k = b2k; b1s = b2s; b2k = b3k; b2s = b3s; b3k = 0;
    }
    // a lane that has handed out all three of its bests may still hold the next one: refill (all lanes, rare)
    const bool empty_but_more = (b1k == 0) && (__popcll(done) < nvalid);
    if (__any(empty_but_more)) refill();
    const int before = ncand;
    rescore(g);
    ++visited;
    if (!have_k) {
      if (visited >= k && ncand >= k) take_top_k();      // first estimate of tau after k groups
    } else if (ncand > before) {
      take_top_k();                                      // something beat tau: tighten it
    }
  }
  take_top_k();
  if (lane < ncand) {
    const uint64_t ck = best[lane];
    out_idx[u * k + lane] = (int32_t)(~(uint32_t)(ck & 0xFFFFFFFFull));
    out_val[u * k + lane] = f32_from_order_key((uint32_t)(ck >> 32));
  }
  for (int i = ncand + lane; i < k; i += 64) {
    out_idx[u * k + i] = -1;
    out_val[u * k + i] = NEG_INF_F;
  }
  if (lane == 0) out_cnt[u] = ncand;
}

// =============================================================================================================
// k <= 16, fused form: the MFMA pass itself keeps, per user, lane half and item slice, the QS_TOPK_K best admissible
// candidates (QM_TOPK epilogue of qstream_kernel: a candidate is looked at only if it beats the running k-th best; seen
// items are dropped when a candidate is merged into the list).  What is left is this merge of the 2 * nsplit short
// sorted lists of a user -- no group maxima in HBM, no transpose, no re-scoring pass.
// One thread per user: k rounds of "largest head".  Keys sort as (score desc, candidate row asc); candidate rows are
// ascending in the global item id, so this is the (score desc, item id asc) order of SURVEY 8.0 S7.
// =============================================================================================================
__global__ __launch_bounds__(256) void topk_merge_kernel(const unsigned long long* __restrict__ keys, int nsplit,
                                                         int64_t n_users, const int32_t* __restrict__ item_ids, int k,
                                                         int32_t* __restrict__ out_idx, float* __restrict__ out_val,
                                                         int32_t* __restrict__ out_cnt) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_users) return;
  const int nl = 2 * nsplit;
  int head[64];                    // nl <= 64 lists (nsplit <= 32)
  for (int l = 0; l < nl; ++l) head[l] = 0;
  int cnt = 0;
  for (int j = 0; j < k; ++j) {
    unsigned long long best = 0ull;
    int bl = -1;
    for (int l = 0; l < nl; ++l) {
      if (head[l] >= QS_TOPK_K) continue;
      const int sp = l >> 1, hh = l & 1;
      const unsigned long long v = keys[(((int64_t)sp * n_users + u) * 2 + hh) * QS_TOPK_K + head[l]];
      if (v > best) {
        best = v;
        bl = l;
      }
    }
    if (bl < 0) break;
    head[bl] += 1;
    const uint32_t c = ~(uint32_t)(best & 0xFFFFFFFFull);
    out_idx[u * k + j] = item_ids ? item_ids[c] : (int32_t)c;
    out_val[u * k + j] = f32_from_order_key((uint32_t)(best >> 32));
    ++cnt;
  }
  for (int j = cnt; j < k; ++j) {
    out_idx[u * k + j] = -1;
    out_val[u * k + j] = NEG_INF_F;
  }
  out_cnt[u] = cnt;
}

// Seen lists -> one bitmap row per user of the chunk (bit c of row u set: candidate row c is excluded for u), so
// that the scoring kernel decides "seen?" with ONE load per candidate that beats its bound.  288 GB of HBM make
// n_users * n_cand / 8 bytes affordable (0.8 GB for 65536 users x 100 000 items); zeroing it is a memset at HBM rate.
// One wave per user; the first entry of every 32-item word ORs in the entries that follow in the same word (lists are
// ascending, so they are adjacent); atomicOr keeps an unsorted list merely slower, not wrong.
__global__ __launch_bounds__(256) void topk_seen_bits_kernel(const int64_t* __restrict__ seen_off,
                                                             const int32_t* __restrict__ seen_items,
                                                             const int32_t* __restrict__ seen_rows, int64_t n_users,
                                                             int64_t n_cand, int64_t W, uint32_t* __restrict__ bits) {
  const int lane = threadIdx.x & 63;
  const int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (u >= n_users) return;
  const int64_t srow = seen_rows ? (int64_t)seen_rows[u] : u;
  const int64_t lo = seen_off[srow], hi = seen_off[srow + 1];
  for (int64_t j = lo + lane; j < hi; j += 64) {
    const int32_t id = seen_items[j];
    if (id < 0 || id >= n_cand) continue;
    const int32_t w = id >> 5;
    if (j > lo && (seen_items[j - 1] >> 5) == w) continue;
    uint32_t m = 1u << (id & 31);
    for (int64_t jj = j + 1; jj < hi; ++jj) {
      const int32_t id2 = seen_items[jj];
      if ((id2 >> 5) != w) break;
      m |= 1u << (id2 & 31);
    }
    atomicOr(&bits[u * W + w], m);
  }
}

// The same merge for at most four lists per user (item slices <= 2, the usual case: one block per CU needs no slicing
// from 65 536 users on), one WAVE per user: lane l holds key l & 15 of list l >> 4; the rank of a key in the union is
// its position in its own list plus, per other list, the number of keys there that beat it -- a 5-step binary search
// through wave shuffles over that list's 16 lanes (lists are sorted, keys distinct).  Keys of rank < k go straight to
// their output slot: no loop over k, no per-thread scratch arrays, coalesced loads (the per-thread form above took
// 70-100 us per 65 536 users, this one ~10).
__global__ __launch_bounds__(256) void topk_merge_wave_kernel(const unsigned long long* __restrict__ keys, int nsplit,
                                                              int64_t n_users, const int32_t* __restrict__ item_ids, int k,
                                                              int32_t* __restrict__ out_idx, float* __restrict__ out_val,
                                                              int32_t* __restrict__ out_cnt) {
  const int lane = threadIdx.x & 63;
  const int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (u >= n_users) return;
  const int nl = 2 * nsplit, lst = lane >> 4, j = lane & 15;
  unsigned long long key = 0ull;
  if (lst < nl) key = keys[(((int64_t)(lst >> 1) * n_users + u) * 2 + (lst & 1)) * QS_TOPK_K + j];
  int rank = j;
  for (int m = 0; m < nl; ++m) {
    // number of keys of list m greater than `key`: binary search over lanes 16 m .. 16 m + 15 (descending order)
    int pos = 0;
#pragma unroll
    for (int step = 8; step >= 1; step >>= 1) {
      const unsigned long long probe = __shfl(key, 16 * m + pos + step - 1);
      if (probe > key) pos += step;
    }
    const unsigned long long last = __shfl(key, 16 * m + pos);      // pos <= 15
    if (last > key) pos += 1;
    if (m != lst) rank += pos;
  }
  const bool take = (lst < nl) && key != 0ull && rank < k;
  const int cnt = __builtin_popcountll(__builtin_amdgcn_ballot_w64(take));
  if (take) {
    const uint32_t c = ~(uint32_t)(key & 0xFFFFFFFFull);
    out_idx[u * k + rank] = item_ids ? item_ids[c] : (int32_t)c;
    out_val[u * k + rank] = f32_from_order_key((uint32_t)(key >> 32));
  }
  if (lane >= cnt && lane < k) {
    out_idx[u * k + lane] = -1;
    out_val[u * k + lane] = NEG_INF_F;
  }
  if (lane == 0) out_cnt[u] = cnt;
}

// launches the merge of the per-(slice, user, lane half) lists
static void tk_launch_merge(const unsigned long long* keys, int nsplit, int64_t n_users, const int32_t* item_ids, int k,
                            int32_t* out_idx, float* out_val, int32_t* out_cnt, hipStream_t s) {
  if (2 * nsplit <= 4)
    hipLaunchKernelGGL(topk_merge_wave_kernel, dim3(cql_ceil_div(n_users, 4)), dim3(256), 0, s, keys, nsplit, n_users,
                       item_ids, k, out_idx, out_val, out_cnt);
  else
    hipLaunchKernelGGL(topk_merge_kernel, dim3(cql_ceil_div(n_users, 256)), dim3(256), 0, s, keys, nsplit, n_users, item_ids,
                       k, out_idx, out_val, out_cnt);
}

static inline int64_t align256(int64_t x) { return (x + 255) / 256 * 256; }
static inline int64_t tk_bits_words(int64_t n_cand) { return ((n_cand + 31) / 32 + 3) / 4 * 4; }
#define TK_FUSED_MAX_SPLIT 32
static QSplit tk_fused_split(int64_t n_cand, int64_t n_users, int d) {
  (void)d;
  QSplit sp = qs_choose_split(n_cand, n_users, 2, QS_TI, 512);      // 2 blocks per CU resident (LDS: ring + buffers)
  if (sp.nsplit > TK_FUSED_MAX_SPLIT) {
    const int64_t units = (n_cand + QS_TI - 1) / QS_TI;
    const int64_t upb = (units + TK_FUSED_MAX_SPLIT - 1) / TK_FUSED_MAX_SPLIT;
    sp.split_rows = upb * QS_TI;
    sp.nsplit = (int)((n_cand + sp.split_rows - 1) / sp.split_rows);
  }
  return sp;
}

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
  const int gstride = (ngroups + 63) / 64 * 64;
  const int64_t two_pass = align256((int64_t)ngroups * n_users * 4) + align256((int64_t)gstride * n_users * 4);
  int64_t fused = align256((int64_t)tk_fused_split(n_cand, n_users, d).nsplit * n_users * 2 * QS_TOPK_K * 8) +
                  align256(n_users * tk_bits_words(n_cand) * 4);
  if (cql_topk2_supported(d, k, n_cand)) {
    int ns;
    int64_t sr;
    cql_topk2_split(n_users, n_cand, &ns, &sr);
    const int64_t f2 = align256((int64_t)ns * n_users * 2 * QS_TOPK_K * 8) + align256(cql_topk2_bits_bytes(n_users, n_cand));
    if (f2 > fused) fused = f2;
  }
  return (two_pass > fused ? two_pass : fused) + 256;
}

extern "C" int cqlrec_score_topk(const uint16_t* H_b, int64_t n_users, const uint16_t* E_b, const float* b,
                                 int64_t n_cand, int32_t d, const int32_t* item_ids, const int64_t* seen_off,
                                 const int32_t* seen_items, const int32_t* seen_rows, int32_t k, void* ws,
                                 int64_t ws_bytes, int32_t* out_idx, float* out_val, int32_t* out_cnt,
                                 cqlrec_stream stream) {
  return cqlrec_score_topk_phase(H_b, n_users, E_b, b, n_cand, d, item_ids, seen_off, seen_items, seen_rows, k, ws, ws_bytes,
                                 out_idx, out_val, out_cnt, CQLREC_TOPK_ALL, stream);
}

// does this shape take the on-chip-selection kernel (whose seen bitmap cqlrec_score_topk_phase can build ahead)?
static bool tk_uses_topk2(int32_t d, int32_t k, int64_t n_cand, const int32_t* item_ids) {
  static const int fused_off = getenv("CQL_TOPK_FUSED") && getenv("CQL_TOPK_FUSED")[0] == '0';
  static const int force_generic0 = getenv("CQL_TOPK_GENERIC") ? 1 : 0;
  static const int tk2_off = getenv("CQL_TOPK2") && getenv("CQL_TOPK2")[0] == '0';
  return !tk2_off && !fused_off && !force_generic0 && item_ids == nullptr && cql_topk2_supported(d, k, n_cand);
}

extern "C" int cqlrec_topk_seen_form(const void* ws, int64_t n_users, int64_t n_cand, int32_t d, int32_t k, int32_t* out,
                                     cqlrec_stream stream) {
  CQL_REQUIRE(ws && out && n_users > 0 && n_cand > 0, "topk_seen_form: bad arguments");
  *out = -1;
  if (!tk_uses_topk2(d, k, n_cand, nullptr)) return CQLREC_OK;
  int ns;
  int64_t sr;
  cql_topk2_split(n_users, n_cand, &ns, &sr);
  const uint32_t* bits = (const uint32_t*)((const char*)ws + align256((int64_t)ns * n_users * 2 * QS_TOPK_K * 8));
  const uint32_t* word = cql_topk2_lists_word(bits, n_users, n_cand);
  if (!word) return CQLREC_OK;
  uint32_t v = 0;
  if (hipMemcpyAsync(&v, word, 4, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
      hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
    cql_set_error("topk_seen_form: copying the word back failed");
    return CQLREC_ERR_HIP;
  }
  *out = v ? 1 : 0;
  return CQLREC_OK;
}

extern "C" int cqlrec_score_topk_phase(const uint16_t* H_b, int64_t n_users, const uint16_t* E_b, const float* b,
                                       int64_t n_cand, int32_t d, const int32_t* item_ids, const int64_t* seen_off,
                                       const int32_t* seen_items, const int32_t* seen_rows, int32_t k, void* ws,
                                       int64_t ws_bytes, int32_t* out_idx, float* out_val, int32_t* out_cnt,
                                       int32_t phase, cqlrec_stream stream) {
  CQL_REQUIRE(phase == CQLREC_TOPK_ALL || phase == CQLREC_TOPK_SEEN || phase == CQLREC_TOPK_SCORE ||
                  phase == CQLREC_TOPK_SEEN_BESIDE, "score_topk: phase=%d", phase);
  if (phase == CQLREC_TOPK_SEEN || phase == CQLREC_TOPK_SEEN_BESIDE) {     // the part that does not depend on the state vectors
    CQL_REQUIRE(ws && n_users > 0 && n_cand > 0, "score_topk (seen phase): bad arguments");
    CQL_REQUIRE(ws_bytes >= cqlrec_topk_ws_bytes(n_users, n_cand, d, k), "score_topk: workspace too small");
    if (!seen_off || !tk_uses_topk2(d, k, n_cand, item_ids)) return CQLREC_OK;   // other forms filter while they select
    CQL_REQUIRE(seen_items != nullptr, "score_topk: seen_items is NULL");
    int ns;
    int64_t sr;
    cql_topk2_split(n_users, n_cand, &ns, &sr);
    uint32_t* bits = (uint32_t*)((char*)ws + align256((int64_t)ns * n_users * 2 * QS_TOPK_K * 8));
    CqlProfScope prof(CQLREC_PH_TOPK_SELECT, (hipStream_t)stream);
    return cql_topk2_seen_bits(seen_off, seen_items, seen_rows, n_users, n_cand, bits, (hipStream_t)stream,
                               phase == CQLREC_TOPK_SEEN_BESIDE);
  }
  CQL_REQUIRE(H_b && E_b && b && ws && out_idx && out_val && out_cnt, "score_topk: NULL pointer");
  CQL_REQUIRE(d == 64 || d == 128 || d == 256, "score_topk: d=%d unsupported", d);
  CQL_REQUIRE(n_users > 0 && n_cand > 0, "score_topk: n_users=%lld n_cand=%lld", (long long)n_users, (long long)n_cand);
  CQL_REQUIRE(k > 0 && k <= TK_MAX_K, "score_topk: k=%d out of range (1..%d)", k, TK_MAX_K);
  CQL_REQUIRE(seen_off == nullptr || seen_items != nullptr, "score_topk: seen_items is NULL");
  CQL_REQUIRE(ws_bytes >= cqlrec_topk_ws_bytes(n_users, n_cand, d, k), "score_topk: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  static const int fused_off = getenv("CQL_TOPK_FUSED") && getenv("CQL_TOPK_FUSED")[0] == '0';   // A/B knob; tests run both
  static const int force_generic0 = getenv("CQL_TOPK_GENERIC") ? 1 : 0;
  // d = 128: one wave per SIMD, selection on chip (qhead_topk2.hip)
  if (tk_uses_topk2(d, k, n_cand, item_ids)) {
    QTk2Args a2 = {};
    a2.H_b = H_b;
    a2.n_users = n_users;
    a2.E_b = E_b;
    a2.bias = b;
    a2.n_cand = n_cand;
    a2.k = k;
    cql_topk2_split(n_users, n_cand, &a2.nsplit, &a2.split_rows);
    a2.keys = (unsigned long long*)ws;
    if (seen_off) {
      uint32_t* bits = (uint32_t*)((char*)ws + align256((int64_t)a2.nsplit * n_users * 2 * QS_TOPK_K * 8));
      if (phase == CQLREC_TOPK_ALL) {
        CqlProfScope prof(CQLREC_PH_TOPK_SELECT, s);
        const int rc = cql_topk2_seen_bits(seen_off, seen_items, seen_rows, n_users, n_cand, bits, s);
        if (rc != CQLREC_OK) return rc;
      }
      a2.seen_bits = bits;
      a2.guard = cql_topk2_lists_word(bits, n_users, n_cand);
      a2.seen_lists = a2.guard ? bits : nullptr;          // (the lists live in the bitmap's space)
    }
    {
      CqlProfScope prof(CQLREC_PH_TOPK_TILEMAX, s);
      const int rc = cql_topk2_run(a2, d, s);
      if (rc != CQLREC_OK) return rc;
    }
    CqlProfScope prof(CQLREC_PH_TOPK_SELECT, s);
    tk_launch_merge((const unsigned long long*)ws, a2.nsplit, n_users, item_ids, (int)k, out_idx, out_val, out_cnt, s);
    CQL_LAUNCH_CHECK("score_topk (topk2)");
    return CQLREC_OK;
  }
  // candidate subsets (item_ids) keep the two-pass form: the bitmap is indexed by candidate row = global item id
  if (k <= QS_TOPK_K && !fused_off && !force_generic0 && item_ids == nullptr && n_cand < (1ll << 31)) {
    const QSplit sp = tk_fused_split(n_cand, n_users, d);
    uint32_t* bits = nullptr;
    const int64_t W = tk_bits_words(n_cand);
    if (seen_off) {
      bits = (uint32_t*)((char*)ws + align256((int64_t)sp.nsplit * n_users * 2 * QS_TOPK_K * 8));
      CqlProfScope prof(CQLREC_PH_TOPK_SELECT, s);
      if (hipMemsetAsync(bits, 0, (size_t)n_users * W * 4, s) != hipSuccess) {
        cql_set_error("score_topk: hipMemsetAsync failed");
        return CQLREC_ERR_HIP;
      }
      hipLaunchKernelGGL(topk_seen_bits_kernel, dim3(cql_ceil_div(n_users, 4)), dim3(256), 0, s, seen_off, seen_items,
                         seen_rows, n_users, n_cand, W, bits);
    }
    QArgs a = {};
    a.res = H_b;
    a.n_res = n_users;
    a.str = E_b;
    a.n_str = n_cand;
    a.str_scalar = b;
    a.nsplit = sp.nsplit;
    a.split_rows = sp.split_rows;
    a.tg = 1;
    a.topk_keys = (unsigned long long*)ws;
    a.topk_k = k;
    a.seen_bits = bits;
    a.seen_w = W;
    qs_launch(k <= 10 ? QM_TOPK10 : QM_TOPK, a, d, sp.rblks, s);      // k <= 10: only 10 list entries are kept sorted
    CqlProfScope prof(CQLREC_PH_TOPK_SELECT, s);
    tk_launch_merge((const unsigned long long*)ws, sp.nsplit, n_users, item_ids, (int)k, out_idx, out_val, out_cnt, s);
    CQL_LAUNCH_CHECK("score_topk (fused)");
    return CQLREC_OK;
  }
  int tg;
  const int ngroups = tk_tile_groups(n_cand, &tg);
  const int gstride = (ngroups + 63) / 64 * 64;
  float* tm = (float*)ws;
  float* tm_t = (float*)((char*)ws + align256((int64_t)ngroups * n_users * 4));
  // pass 1
  const int unit = (32 * tg > QS_TI) ? 32 * tg : QS_TI;
  const QSplit sp = qs_choose_split(n_cand, n_users, qs_spw_fwd(d), unit, QS_TARGET_BLOCKS);
  QArgs a = {};
  a.res = H_b;
  a.n_res = n_users;
  a.str = E_b;
  a.n_str = n_cand;
  a.str_scalar = b;
  a.nsplit = sp.nsplit;
  a.split_rows = sp.split_rows;
  a.tilemax = tm;
  a.tg = tg;
  qs_launch(QM_TILEMAX, a, d, sp.rblks, s);
  // pass 2
  CqlProfScope prof(CQLREC_PH_TOPK_SELECT, s);
  hipLaunchKernelGGL(tilemax_transpose_kernel, dim3(cql_ceil_div(n_users, 32), cql_ceil_div(gstride, 32)), dim3(256), 0,
                     s, tm, ngroups, n_users, tm_t, gstride);
  dim3 grid((unsigned)n_users), block(64);
#define TK_LAUNCH_CB(DD, KP, CB)                                                                                      \
  hipLaunchKernelGGL((topk_select_kernel<DD, KP, CB>), grid, block, 0, s, H_b, n_users, E_b, b, n_cand, item_ids,        \
                     seen_off, seen_items, seen_rows, (const float*)tm_t, gstride, ngroups, tg, k, out_idx, out_val,     \
                     out_cnt)
#define TK_LAUNCH(DD, KP)                                     \
  do {                                                        \
    if (k <= TK_CB_SMALL / 2) TK_LAUNCH_CB(DD, KP, TK_CB_SMALL); \
    else TK_LAUNCH_CB(DD, KP, TK_CB_LARGE);                   \
  } while (0)
#define TKS_LAUNCH(DD, KP)                                                                                          \
  hipLaunchKernelGGL((topk_select_small_kernel<DD, KP>), grid, block, 0, s, H_b, n_users, E_b, b, n_cand, item_ids,  \
                     seen_off, seen_items, seen_rows, (const float*)tm_t, gstride, ngroups, tg, k, out_idx, out_val,  \
                     out_cnt)
  static const int force_generic = getenv("CQL_TOPK_GENERIC") ? 1 : 0;   // tests run both kernels
  const bool small_k = (k <= TKS_MAX_K) && !force_generic;
#define TK_BY_KPL(DD)                                                       \
  do {                                                                      \
    if (ngroups <= 1024) { if (small_k) TKS_LAUNCH(DD, 16); else TK_LAUNCH(DD, 16); }      \
    else if (ngroups <= 2048) { if (small_k) TKS_LAUNCH(DD, 32); else TK_LAUNCH(DD, 32); } \
    else { if (small_k) TKS_LAUNCH(DD, 64); else TK_LAUNCH(DD, 64); }                      \
  } while (0)
  if (d == 64) TK_BY_KPL(64); else if (d == 128) TK_BY_KPL(128); else TK_BY_KPL(256);
#undef TK_BY_KPL
#undef TKS_LAUNCH
#undef TK_LAUNCH
#undef TK_LAUNCH_CB
  CQL_LAUNCH_CHECK("score_topk");
  return CQLREC_OK;
}
