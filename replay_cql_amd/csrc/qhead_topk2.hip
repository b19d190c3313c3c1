// Top-K scoring pass, second form (d = 128): ONE wave per SIMD, each wave owns TWO 32-user groups (64 users, 256 per
// block), the item catalogue streams through an LDS ring filled by LDS-DMA, and the k best admissible items of every
// user are kept ON CHIP -- no score, no tile maximum and no candidate leaves the CU before the pass is over.
// Replaces, for k <= 16 and whole-catalogue scoring, both the QM_TILEMAX + select pair and the QM_TOPK mode of
// qstream_kernel (two waves per SIMD, lists in registers, bitmap look-ups in global memory: 3.2 ms for 65536 users x
// 100 000 items, MFMA busy 20 %; its waves stall each other at the stage barrier whenever one of them merges).
//
// What the reference does here: CQL._predict scores every (user, item) pair with the Q network, drops seen items
// (replay/models/base_rec.py:417-464 _filter_seen), keeps the k largest per user (base_rec.py get_top_k / :684-740).
//
// Per tile (32 items) and wave: two MFMA chains of 8 (one per user group; A = item rows from LDS, B = user fragments
// resident in registers, C of the first product = the items' bias) and, behind each chain, its epilogue:
//
//   fast path (every tile)   4 quad maxima + tile maximum of the lane's 16 scores (10 VALU, spread over the gaps of
//                            the FOLLOWING chain), one compare against the lane's bound, one wave-wide branch
//   slow path (some lane has a score >= its bound)   the quads, then the elements of a quad that beat the bound:
//                            seen? (one bit of a word that came in with the stage) -> append a 64-bit key to the
//                            lane's private queue in LDS (8 entries)
//   merge (a queue is full; ~25 times per pass and group)   the lane's sorted list of 16 keys (AGPRs) takes the queued
//                            keys in; the new bound is the k-th key of the list or of the partner lane's list
//                            (lane ^ 32 holds the same user's other 16 rows of every tile), whichever is larger
//
// Keys sort as (score desc, item row asc): order-preserving score bits << 32 | ~row.  The bound of a lane is always
// the k-th best ADMISSIBLE score already known for its user, so nothing that belongs to the user's top k is ever
// dropped (ties with the bound are kept: ">=").  Per (item slice, user, lane half) the list goes to HBM at the end and
// topk_merge_kernel (topk.hip) merges the 2 * nsplit lists of a user.
//
// Seen items: a bitmap built once per call, laid out the way the pass consumes it -- [64-user block][64-item stage]
// [user][2 words] -- so that a wave's 512 bytes of a stage are one contiguous piece of the stage's LDS-DMA.
//
// LDS (one block per CU): ring 3 x (16 KiB rows + 4 x 256 B bias strips + 4 x 512 B seen words) = 57 KiB,
// queues 4 waves x 2 groups x 24 x 512 B = 96 KiB.  The lists (2 x 16 keys per lane) sit in AGPRs.
#include <stdlib.h>
#include <type_traits>
#include "qhead_de_common.h"

#define TK2_Q 24         // queue entries per lane and group: a tile adds at most 16, so a merge in front of a tile
                         // whenever some lane holds more than 8 keeps the scan of the tile free of overflow handling
#define TK2_K 16         // list entries (= QS_TOPK_K: the format topk_merge_kernel reads)
#define TK2_NBUF 3        // ring buffers (a deeper ring, 6 x 19 KB, was measured: no faster -- the stage is not latency bound)
static_assert(TK2_K == QS_TOPK_K, "list format shared with topk_merge_kernel");

typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((address_space(3))) u32x2 lds_u2;

template <int D>
struct Tk2Cfg {
  using C = DeCfg<D, 4>;
  static constexpr int STRIPS = 4 * 256;                 // one private bias strip (64 floats) per wave
  static constexpr int SEEN = 4 * 512;                   // per wave: 64 users x 2 words
  static constexpr int BUF = C::STAGE_BYTES + STRIPS + SEEN;
  static constexpr int RING = TK2_NBUF * BUF;
  static constexpr int QUEUES = 4 * 2 * TK2_Q * 512;
  static constexpr int SMEM = RING + QUEUES;
  static constexpr int VPS = C::LPS + 2;                 // LDS-DMA instructions per wave and stage
};

// at most `stages` younger stages (6 LDS-DMA instructions each) may still be in flight
__device__ __forceinline__ void tk2_wait_stages(int stages) {
  if (stages <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (stages == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (stages == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (stages == 3) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
  else if (stages == 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
}
// the 32 low lanes move 512 B (a wave's seen words of a stage) by LDS-DMA; the others stay out of it
__device__ __forceinline__ void bdma16_lo32(uint32_t voff, __amdgpu_buffer_rsrc_t rsrc, uint32_t soff, uint32_t lds_dst) {
  uint32_t keep;
  uint64_t ex;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %5\n\ts_mov_b64 exec, 0xffffffff\n\t"
               "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep), "=&s"(ex) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}
static_assert(TK2_NBUF <= 7, "tk2_wait_stages covers up to five younger stages; vmcnt counts to 63");
// the sorted lists live in the AGPR half of the register file (one wave per SIMD: 256 of them, otherwise idle) and come
// down to VGPRs only inside a merge
#define TK2_AW(dst, src) asm("v_accvgpr_write_b32 %0, %1" : "=a"(dst) : "v"(src))
#define TK2_AR(dst, src) asm("v_accvgpr_read_b32 %0, %1" : "=v"(dst) : "a"(src))

// max of four without the canonicalising self-maxima hipcc puts in front of fmaxf (scores are never NaN: finite inputs)
__device__ __forceinline__ float tk2_max4(float a, float b, float c, float d) {
  float t, q;
  asm("v_max_f32 %0, %1, %2" : "=v"(t) : "v"(c), "v"(d));
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(q) : "v"(a), "v"(b), "v"(t));
  return q;
}

__device__ __forceinline__ float tk2_max3(float a, float b, float c) {
  float q;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(q) : "v"(a), "v"(b), "v"(c));
  return q;
}

// KC: list entries that are kept sorted (10 for k <= 10, else 16; see qtopk4_kernel): what falls off position KC - 1 of a
// (slice, user, lane half) list cannot be among that list's best k, and an insertion costs ~8 instructions per key passed.
template <int D, int KC>
__global__ __launch_bounds__(256, 1) void qtopk2_kernel(QTk2Args a) {
  static_assert(KC == 10 || KC == 16, "lists of 10 or 16 sorted keys");
  using T = Tk2Cfg<D>;
  using C = typename T::C;
  constexpr int KS = C::KS;
  static_assert(C::TILES == 2 && T::VPS == 6, "two tiles per stage, six pieces per wave (tk2_wait_stages)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // the ONLY LDS object of this kernel

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int split = blockIdx.x % a.nsplit;
  const int64_t rblk = blockIdx.x / a.nsplit;
  const int64_t res0 = rblk * 256 + wave * 64;
  const int64_t s_begin = (int64_t)split * a.split_rows;
  const int64_t s_end = (s_begin + a.split_rows < a.n_cand) ? (s_begin + a.split_rows) : a.n_cand;
  const int nstage = (s_end > s_begin) ? (int)((s_end - s_begin + C::TI - 1) / C::TI) : 0;
  const uint32_t gstage0 = (uint32_t)(s_begin / C::TI);

  // ---- user fragments ------------------------------------------------------------------------------------------------
  bf16x8 rf[2][KS];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    int64_t row = res0 + g * 32 + r;
    if (row >= a.n_users) row = a.n_users - 1;
#pragma unroll
    for (int s = 0; s < KS; ++s) rf[g][s] = *reinterpret_cast<const bf16x8*>(a.H_b + row * D + 16 * s + 8 * h);
  }
  // retired in hipcc's own bookkeeping before the first LDS-DMA is issued (see qde2_kernel::load_owner)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- staging ---------------------------------------------------------------------------------------------------------
  __amdgpu_buffer_rsrc_t rs_e = __builtin_amdgcn_make_buffer_rsrc((void*)a.E_b, 0, (int)(a.n_cand * C::ROWB), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, (int)(a.n_cand * 4), 0x00020000);
  // seen words of this wave's 64 users: [stage][64 users][2 words], 512 B per stage; no filter = an empty buffer (reads 0)
  // A wave whose 64 users all lie past n_users (the tail of the last 256-user row-block) has no group in the bitmap --
  // it holds ceil(n_users / 64) groups -- and gets the empty buffer too: its rows are discarded, and num_records is
  // relative to wsrc, so the hardware bounds check would not stop the reads behind the bitmap's end.
  const int64_t nst_all = (a.n_cand + C::TI - 1) / C::TI;
  const bool has_seen = a.seen_bits != nullptr && res0 < a.n_users;        // wave-uniform
  const uint32_t* wsrc = has_seen ? a.seen_bits + (res0 >> 6) * nst_all * 128 : (const uint32_t*)a.bias;
  __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc((void*)wsrc, 0, has_seen ? (int)(nst_all * 512) : 0, 0x00020000);
  uint32_t voff;
  {
    const int sub = lane >> 5, r7 = (lane >> 2) & 7, slot = lane & 3;
    const int rg0 = wave / C::PPG, hc = wave % C::PPG;
    const int q2 = (r7 >> 2) | ((rg0 & 1) << 1);
    voff = (uint32_t)((rg0 * 8 + r7) * C::ROWB + (8 * hc + 4 * sub + (slot ^ q2)) * 16);
  }
  const uint32_t voff4 = (uint32_t)lane * 4;
  const uint32_t smem_base = lds_addr_of(smem);
  // piece `pc` (0..VPS-1) of stage `stage` into ring buffer `buf`
  auto issue_piece = [&](int stage, int buf, int pc) __attribute__((always_inline)) {
    const uint32_t bufp = __builtin_amdgcn_readfirstlane(smem_base + buf * T::BUF);
    const uint32_t gs = gstage0 + (uint32_t)stage;
    if (pc < C::LPS) bdma16(voff, rs_e, gs * C::STAGE_BYTES + C::PSTEP * pc, bufp + (4 * pc + wave) * 1024);
    else if (pc == C::LPS) bdma4(voff4, rs_b, gs * (C::TI * 4), bufp + C::STAGE_BYTES + wave * 256);
    else bdma16_lo32(4 * voff4, rs_w, gs * 512, bufp + C::STAGE_BYTES + T::STRIPS + wave * 512);
  };
  auto issue = [&](int stage, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int pc = 0; pc < T::VPS; ++pc) issue_piece(stage, buf, pc);
  };

  // ---- read geometry: per-lane offsets inside a ring buffer (qde_kernel's image); the buffer offsets rotate ----------
  const lds_u8* lbase = (const lds_u8*)smem;
  const int oa0 = C::RG_BYTES * (r >> 3) + 64 * (r & 7) + 16 * ((0 + h) ^ ((r >> 2) & 3));
  const int oa1 = C::RG_BYTES * (r >> 3) + 64 * (r & 7) + 16 * ((2 + h) ^ ((r >> 2) & 3));
  const int os = C::STAGE_BYTES + wave * 256 + 16 * h;
  const int ow = C::STAGE_BYTES + T::STRIPS + wave * 512 + r * 8;      // + 256 for the second user group
  const lds_u8 *pA0, *pA1, *pS, *pW;       // current buffer
  const lds_u8 *nA0, *nA1, *nS;            // next buffer
  int boff_c = 0, boff_n = T::BUF;
  auto set_ptrs = [&]() __attribute__((always_inline)) {
    pA0 = lbase + boff_c + oa0; pA1 = lbase + boff_c + oa1; pS = lbase + boff_c + os; pW = lbase + boff_c + ow;
    nA0 = lbase + boff_n + oa0; nA1 = lbase + boff_n + oa1; nS = lbase + boff_n + os;
  };
  set_ptrs();

  // ---- per-lane selection state ------------------------------------------------------------------------------------
  lds_u8* qb[2];            // queue of group g: entry e at + 512 e
  uint32_t la[2][2 * TK2_K];   // list of group g: key j = la[g][2j] | la[g][2j+1] << 32   (AGPRs)
  float thr[2];
  int cnt[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    qb[g] = (lds_u8*)smem + T::RING + ((wave * 2 + g) * TK2_Q * 64 + lane) * 8;
#ifdef TK2_ABL_NOSLOW       // timing-only build: nothing ever beats the bound
    thr[g] = 3.0e38f;
#else
    thr[g] = -3.0e38f;      // finite: an item that scores -inf is never selected
#endif
    cnt[g] = 0;
#pragma unroll
    for (int j = 0; j < 2 * TK2_K; ++j) TK2_AW(la[g][j], 0u);
  }
  const int kth = a.k - 1;
  const bool kb0 = kth & 1, kb1 = kth & 2, kb2 = kth & 4, kb3 = kth & 8;

  // the queue of group g goes into its list; new bound
  auto merge = [&](auto G_) __attribute__((always_inline)) {
    constexpr int g = decltype(G_)::value;
    unsigned long long lst[TK2_K];
#pragma unroll
    for (int j = 0; j < TK2_K; ++j) {
      if (j < KC) {
        uint32_t lo, hi;
        TK2_AR(lo, la[g][2 * j]);
        TK2_AR(hi, la[g][2 * j + 1]);
        lst[j] = ((unsigned long long)hi << 32) | lo;
      } else {
        lst[j] = 0ull;         // (never holds a key: the registers behind it stay zero from the start)
      }
    }
#pragma unroll 1
    for (int e = 0; __builtin_amdgcn_ballot_w64(e < cnt[g]) != 0; ++e) {
      unsigned long long kx = (e < cnt[g]) ? *(const lds_u64*)(qb[g] + 512 * e) : 0ull;
#pragma unroll
      for (int j = 0; j < KC; ++j) {       // insertion into the sorted list: keys are distinct
        const bool gt = kx > lst[j];
        const unsigned long long hi_ = gt ? kx : lst[j];
        kx = gt ? lst[j] : kx;
        lst[j] = hi_;
      }
    }
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      TK2_AW(la[g][2 * j], (uint32_t)lst[j]);
      TK2_AW(la[g][2 * j + 1], (uint32_t)(lst[j] >> 32));
    }
    cnt[g] = 0;
    // the k-th key: a select tree over the bits of k - 1, written with bit masks (v_bfi): as "c ? lst[2j+1] : lst[2j]"
    // hipcc turns it into a run-time index and the list into a scratch array
    const unsigned long long m0 = kb0 ? ~0ull : 0ull, m1 = kb1 ? ~0ull : 0ull, m2 = kb2 ? ~0ull : 0ull,
                             m3 = kb3 ? ~0ull : 0ull;
    unsigned long long t8[8], t4[4], t2[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) t8[j] = (lst[2 * j + 1] & m0) | (lst[2 * j] & ~m0);
#pragma unroll
    for (int j = 0; j < 4; ++j) t4[j] = (t8[2 * j + 1] & m1) | (t8[2 * j] & ~m1);
#pragma unroll
    for (int j = 0; j < 2; ++j) t2[j] = (t4[2 * j + 1] & m2) | (t4[2 * j] & ~m2);
    const unsigned long long kk = (t2[1] & m3) | (t2[0] & ~m3);
    // A later item of THIS lane has a larger row than every key of this list, so it must beat the list's k-th score
    // strictly; against the partner lane's k-th score (rows of the other half of every tile) a tie still counts.
    const uint32_t okey = (uint32_t)(kk >> 32);
    const float own = (kk != 0ull) ? f32_from_order_key(okey) : -3.0e38f;
    const float own_up = (kk != 0ull) ? f32_from_order_key(okey + 1u) : -3.0e38f;
    thr[g] = fmaxf(own_up, __shfl_xor(own, 32));
  };

  // slow path of the epilogue of one (tile, group): trow0 = global row of the tile's row 0, w = seen bits of the tile's
  // 32 rows for this lane's user.  One merge site in front of the scan (see TK2_Q); the scan itself is straight-line.
  auto slow = [&](const f32x16& acc, auto G_, int64_t trow0, uint32_t w) __attribute__((always_inline)) {
    constexpr int g = decltype(G_)::value;
    if (__builtin_amdgcn_ballot_w64(cnt[g] > TK2_Q - 16) != 0) merge(G_);
    float qm[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) qm[q] = tk2_max4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    const uint32_t wh = w >> (4 * h);
    const int lim = (int)((s_end - trow0 < 64) ? (s_end - trow0) : 64) - 4 * h;   // element admissible iff rc < lim
    const uint32_t nrow = ~((uint32_t)trow0 + 4u * (uint32_t)h);                  // ~(row0 + rc) = nrow - rc
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (__builtin_amdgcn_ballot_w64(qm[q] >= thr[g]) == 0) continue;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int i = 4 * q + jj;
        const int rc = 8 * q + jj;
        const bool c0 = acc[i] >= thr[g];
        if (__builtin_amdgcn_ballot_w64(c0) == 0) continue;
        if (c0) {
          uint32_t whv = wh, nrv = nrow;      // opaque: hipcc would otherwise evaluate all 16 tests, row ids and keys
          int limv = lim;                     // ahead of the scan and hold them in registers
          float av = acc[i];
          asm volatile("" : "+v"(whv), "+v"(nrv), "+v"(limv), "+v"(av));
          if ((((whv >> rc) & 1u) == 0u) && (rc < limv)) {
            const unsigned long long key =
                ((unsigned long long)f32_order_key(av) << 32) | (unsigned long long)(nrv - (uint32_t)rc);
            *(lds_u64*)(qb[g] + 512 * cnt[g]) = key;
            cnt[g] += 1;
          }
        }
      }
    }
  };

#define TK2_FENCE() __builtin_amdgcn_sched_barrier(0)
  bf16x8 af[2][KS];        // item-row fragments and bias by tile parity: those of the next tile are read, one per gap, while
  f32x16 sv[2];            // the chains of this tile run (all four waves of a CU read in step: 13 reads inside ONE chain
  f32x16 acc0, acc1;       // of 8 keep the LDS at ~90 % for that chain, and the chain waits for its fragments)
#pragma unroll
  for (int i = 0; i < 16; ++i) acc1[i] = NEG_INF_F;       // "the tile before the first": nothing beats any bound
  u32x2 wv0 = {0u, 0u}, wv1 = {0u, 0u};   // seen words of the stage: .x tile 0, .y tile 1
  uint32_t w_pend = 0u;
  float qm[4], tmx = 0.f;      // scratch of the fast path (qm: partial maxima)

  // LDS reads of the tile FOLLOWING tile IT of the current buffer: idx 0..7 rows, 8..11 bias
  auto next_read = [&](auto IT, int idx) __attribute__((always_inline)) {
    constexpr bool END = decltype(IT)::value == 1;
    constexpr int NIT = END ? 0 : 1;
    constexpr int noff = NIT * C::TILE_BYTES;
    if (idx < KS) {
      af[NIT][idx] = *(const lds_bf16x8*)(((idx & 1) ? (END ? nA1 : pA1) : (END ? nA0 : pA0)) + noff + 512 * (idx >> 1));
    } else {
      const int q = idx - KS;
      const f32x4 t4 = *(const lds_f4*)((END ? nS : pS) + 128 * NIT + 32 * q);
      sv[NIT][4 * q + 0] = t4[0];
      sv[NIT][4 * q + 1] = t4[1];
      sv[NIT][4 * q + 2] = t4[2];
      sv[NIT][4 * q + 3] = t4[3];
    }
  };
  // epilogue pieces of a finished chain, spread over gaps 3..6 of the chain that follows it (the accumulator is
  // complete three gaps after its last product was issued, and must be read before the chain after next overwrites it)
  auto epi = [&](int gp, const f32x16& acc, auto G_, int64_t trow0, uint32_t w) __attribute__((always_inline)) {
    constexpr int g = decltype(G_)::value;
#ifdef TK2_ABL_NOEPI
    if (gp == 6) asm volatile("" :: "v"(acc[0]), "v"(acc[15]));
    return;
#endif
    // tile maximum of the lane's 16 scores as a v_max3 tree: 8 instructions (the quad maxima the slow path scans by are
    // formed there, not here)
    if (gp == 3) {
      qm[0] = tk2_max3(acc[0], acc[1], acc[2]);
      qm[1] = tk2_max3(acc[3], acc[4], acc[5]);
      qm[2] = tk2_max3(acc[6], acc[7], acc[8]);
    } else if (gp == 4) {
      qm[3] = tk2_max3(acc[9], acc[10], acc[11]);
      tmx = tk2_max3(acc[12], acc[13], acc[14]);
    } else if (gp == 5) {
      qm[0] = tk2_max3(qm[0], qm[1], qm[2]);
      tmx = tk2_max3(qm[3], tmx, acc[15]);
      asm("v_max_f32 %0, %1, %2" : "=v"(tmx) : "v"(qm[0]), "v"(tmx));
    } else if (gp == 6) {
#ifndef TK2_ABL_NOBRANCH
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(tmx >= thr[g]) != 0, 0)) slow(acc, G_, trow0, w);
#else
      asm volatile("" :: "v"(tmx));
#endif
    }
  };

  // the ring turns in front of the reads of the next stage's first tile: that stage has landed for everyone, everyone
  // has left the current stage's buffer (its last reads were issued a period ago; the wait below is for the seen words
  // read in between), which is refilled three stages ahead
  int st = 0, buf_c = 0;
  bool refill = false;
  auto ring_turn = [&]() __attribute__((always_inline)) {
#ifdef TK2_ABL_NOTURN
    if (st < -1) {
#else
    if (st + 1 < nstage) {
#endif
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
      const int last = (nstage - 1 < st + TK2_NBUF - 1) ? nstage - 1 : st + TK2_NBUF - 1;   // youngest stage issued
#ifndef TK2_ABL_NOWAIT
      tk2_wait_stages(last - (st + 1));
#endif
#ifndef TK2_ABL_NOBARRIER
      __builtin_amdgcn_s_barrier();
#endif
    }
#ifdef TK2_ABL_NODMA
    refill = false;
#else
    refill = st + TK2_NBUF < nstage;
#endif
  };

  auto period = [&](auto IT, int64_t row0, int64_t row0_prev) __attribute__((always_inline)) {
    constexpr int P = decltype(IT)::value;
    constexpr bool END = P == 1;
    const uint32_t wE1 = END ? wv1[0] : w_pend;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (END && s == 0) ring_turn();      // in front of the first read of the next stage's buffer
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[P][s], rf[0][s], s == 0 ? sv[P] : acc0, 0, 0, 0);
      TK2_FENCE();
      epi(s, acc1, std::integral_constant<int, 1>{}, row0_prev, wE1);
      TK2_FENCE();
#ifndef TK2_ABL_NOREAD
      next_read(IT, s);
#endif
      TK2_FENCE();
    }
    if constexpr (!END) {      // the new stage's seen words (the pending one of the old stage has just been used)
      wv0 = *(const lds_u2*)pW;
      wv1 = *(const lds_u2*)(pW + 256);
    }
    const uint32_t wE0 = END ? wv0[1] : wv0[0];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[P][s], rf[1][s], s == 0 ? sv[P] : acc1, 0, 0, 0);
      TK2_FENCE();
      // the buffer just left is refilled three stages ahead, one piece per gap (an LDS-DMA holds the wave's issue for
      // ~60 cycles: behind an MFMA half of that is hidden, in a burst at the turn none of it)
      if (END && s < T::VPS && refill) issue_piece(st + TK2_NBUF, buf_c, s);
      TK2_FENCE();
      epi(s, acc0, std::integral_constant<int, 0>{}, row0, wE0);
      TK2_FENCE();
#ifndef TK2_ABL_NOREAD
      if (s < 4) next_read(IT, KS + s);
#endif
      TK2_FENCE();
    }
  };

  if (nstage > 0) {
    // ---- prologue: the whole ring in flight; rows and bias of the first tile in registers -------------------------
    for (int s0 = 0; s0 < TK2_NBUF && s0 < nstage; ++s0) issue(s0, s0);
    tk2_wait_stages(((nstage < TK2_NBUF) ? nstage : TK2_NBUF) - 1);
    __builtin_amdgcn_s_barrier();
    {   // "the tile following the last tile of the buffer before buffer 0": next_read with the buffers' roles swapped
      const lds_u8 *kA0 = nA0, *kA1 = nA1, *kS = nS;
      nA0 = pA0; nA1 = pA1; nS = pS;
#pragma unroll
      for (int idx = 0; idx < KS + 4; ++idx) next_read(std::integral_constant<int, 1>{}, idx);
      nA0 = kA0; nA1 = kA1; nS = kS;
    }
    int64_t row_prev = s_begin;     // (unused by the first period: acc1 = -inf)
    for (st = 0; st < nstage; ++st) {
      const int64_t row0 = s_begin + (int64_t)st * C::TI;
      period(std::integral_constant<int, 0>{}, row0, row_prev);
      period(std::integral_constant<int, 1>{}, row0 + 32, row0);
      row_prev = row0 + 32;
      w_pend = wv1[1];
      // the ring turned: the next buffer is the current one now
      boff_c = boff_n;
      boff_n = (boff_n + T::BUF == T::RING) ? 0 : boff_n + T::BUF;
      buf_c = (buf_c + 1 == TK2_NBUF) ? 0 : buf_c + 1;
      set_ptrs();
    }
    // the last chain's epilogue
#pragma unroll
    for (int gp = 3; gp <= 6; ++gp) epi(gp, acc1, std::integral_constant<int, 1>{}, row_prev, w_pend);
  }

  // ---- flush the queues, write the lists ---------------------------------------------------------------------------
  merge(std::integral_constant<int, 0>{});
  merge(std::integral_constant<int, 1>{});
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int64_t row = res0 + g * 32 + r;
    if (row < a.n_users) {
      unsigned long long* dst = a.keys + (((int64_t)split * a.n_users + row) * 2 + h) * TK2_K;
#pragma unroll
      for (int j = 0; j < TK2_K; j += 2) {
        uint32_t w0, w1, w2, w3;
        TK2_AR(w0, la[g][2 * j]);
        TK2_AR(w1, la[g][2 * j + 1]);
        TK2_AR(w2, la[g][2 * j + 2]);
        TK2_AR(w3, la[g][2 * j + 3]);
        *reinterpret_cast<uint4*>(dst + j) = make_uint4(w0, w1, w2, w3);
      }
    }
  }
}

// =============================================================================================================
// seen lists -> bitmap in the layout above, written DENSELY: one block per 64-user block, the bitmap of TK2_BCH stages at
// a time is assembled in LDS (zero, OR the users' entries of that item range in, write out with 16-byte stores) -- every
// byte of the bitmap is written exactly once, nothing is zeroed beforehand and no read-modify-write reaches HBM
// (memset + atomicOr per entry took 0.30 ms for 65536 users x 100 000 items; the bitmap alone is 0.13 ms of writes).
// The lists are ASCENDING (cqlrec.h): four threads per user walk a list with one cursor per user.
// =============================================================================================================
// BCH = stages per LDS tile (512 B each): 96 (48 KiB) when the builder has the chip to itself; 12 (6 KiB) when it runs on
// a side stream beside qtopk2_kernel, whose block leaves 7 KiB of a CU's LDS free -- with the large tile no block of the
// builder becomes resident before the scoring kernel has finished.
template <int TK2_BCH>
__global__ __launch_bounds__(256) void topk2_seen_bits_kernel(const int64_t* __restrict__ seen_off,
                                                              const int32_t* __restrict__ seen_items,
                                                              const int32_t* __restrict__ seen_rows, int64_t n_users,
                                                              int64_t n_cand, int64_t nst_all, uint32_t* __restrict__ bits,
                                                              const uint32_t* __restrict__ only_if) {
  if (only_if != nullptr && *only_if == 0u) return;       // the entry lists hold everything (qhead_topk4.hip): no bitmap
  __shared__ __attribute__((aligned(16))) uint32_t tile[TK2_BCH * 128];     // [stage][user][2 words]
  const int t = threadIdx.x, user = t >> 2, q = t & 3;
  const int64_t u = (int64_t)blockIdx.x * 64 + user;
  int64_t cur = 0, end = 0;                 // this user's entries not yet placed (kept in step by its four threads)
  if (u < n_users) {
    const int64_t srow = seen_rows ? (int64_t)seen_rows[u] : u;
    cur = seen_off[srow];
    end = seen_off[srow + 1];
  }
  uint32_t* dst = bits + (int64_t)blockIdx.x * nst_all * 128;
  for (int64_t st0 = 0; st0 < nst_all; st0 += TK2_BCH) {
    const int nst = (int)((nst_all - st0 < TK2_BCH) ? (nst_all - st0) : TK2_BCH);
    for (int i = t; i < nst * 32; i += 256) reinterpret_cast<uint4*>(tile)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    const int64_t item_lo = st0 * 64, item_hi = (st0 + nst) * 64;
    int64_t j = cur + q;
    for (; j < end; j += 4) {
      const int32_t id = seen_items[j];
      if (id >= item_hi) break;
      if (id >= item_lo && id < n_cand)
        atomicOr(&tile[(int)((id >> 6) - st0) * 128 + user * 2 + ((id >> 5) & 1)], 1u << (id & 31));
    }
    if (j > end) j = end;
    // first entry at or beyond item_hi = the smallest of the four stopping points (ascending list, stride 4)
    int64_t b = j;
    {
      const int64_t o1 = __shfl_xor(b, 1, 4);
      b = (o1 < b) ? o1 : b;
      const int64_t o2 = __shfl_xor(b, 2, 4);
      b = (o2 < b) ? o2 : b;
    }
    cur = b;
    __syncthreads();
    uint4* out = reinterpret_cast<uint4*>(dst + st0 * 128);
    for (int i = t; i < nst * 32; i += 256) out[i] = reinterpret_cast<const uint4*>(tile)[i];
    __syncthreads();
  }
}

// =============================================================================================================
// host side
// =============================================================================================================
static int tk2_cus() {
  static int cus_dev[CQL_MAX_DEVICES] = {};
  int& cus = cus_dev[cql_device_slot()];
  if (!cus) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  return cus;
}

bool cql_topk2_supported(int d, int k, int64_t n_cand) {
  return d == 128 && k <= TK2_K && n_cand * 256 < (1ll << 31);
}

// item slices: one block per CU when the users alone do not fill the chip
void cql_topk2_split(int64_t n_users, int64_t n_cand, int* nsplit, int64_t* split_rows) {
  const int upb = cql_topk4_use(128, 1, n_users, n_cand) ? 512 : 256;      // users per block (qtopk4_kernel / qtopk2_kernel)
  const int64_t rblks = (n_users + upb - 1) / upb;
  const int64_t stages = (n_cand + 63) / 64;
  int64_t want = (tk2_cus() + rblks - 1) / rblks;
  static const char* env = getenv("CQL_TOPK2_NSPLIT");
  if (env) want = atoi(env);
  if (want > stages / 8) want = stages / 8;      // at least eight stages per slice
  if (want > 16) want = 16;
  if (want < 1) want = 1;
  const int64_t spb = (stages + want - 1) / want;
  *split_rows = spb * 64;
  *nsplit = (int)((n_cand + *split_rows - 1) / *split_rows);
}

static int64_t tk2_dense_bytes(int64_t n_users, int64_t n_cand) {
  return (((n_users + 63) / 64) * ((n_cand + 63) / 64) * 512 + 255) / 256 * 256;
}
static bool tk2_lists(int64_t n_users, int64_t n_cand) {
  return cql_topk4_lists_on() && cql_topk4_use(128, 1, n_users, n_cand) &&
         cql_topk4_lists_fit(n_users, n_cand, tk2_dense_bytes(n_users, n_cand));
}
// the bitmap -- whose space the entry lists of qtopk4_kernel are built in -- and the word that says which of the two it holds
int64_t cql_topk2_bits_bytes(int64_t n_users, int64_t n_cand) {
  return tk2_dense_bytes(n_users, n_cand) + (tk2_lists(n_users, n_cand) ? 256 : 0);
}
const uint32_t* cql_topk2_lists_word(const uint32_t* bits, int64_t n_users, int64_t n_cand) {
  if (!bits || !tk2_lists(n_users, n_cand)) return nullptr;
  return bits + tk2_dense_bytes(n_users, n_cand) / 4;
}

int cql_topk2_seen_bits(const int64_t* seen_off, const int32_t* seen_items, const int32_t* seen_rows, int64_t n_users,
                        int64_t n_cand, uint32_t* bits, hipStream_t s, int beside_scoring) {
  // where the scoring kernel takes entry lists: those first, in the bitmap's space; the bitmap then only if they did not fit
  const uint32_t* only_if = nullptr;
  if (uint32_t* word = const_cast<uint32_t*>(cql_topk2_lists_word(bits, n_users, n_cand))) {
    const int rc = cql_topk4_seen_lists(seen_off, seen_items, seen_rows, n_users, n_cand, bits, tk2_dense_bytes(n_users, n_cand),
                                        word, s);
    if (rc != CQLREC_OK) return rc;
    only_if = word;
  }
  // (as the fall-back behind the lists it returns at once nearly always: the small-tile form, whose 2 048 blocks are
  // handed out in a third of the time -- 8 instead of 23 us on the critical path of every launch)
  if (beside_scoring || only_if != nullptr)
    hipLaunchKernelGGL(topk2_seen_bits_kernel<12>, dim3((unsigned)((n_users + 63) / 64)), dim3(256), 0, s, seen_off,
                       seen_items, seen_rows, n_users, n_cand, (n_cand + 63) / 64, bits, only_if);
  else
    hipLaunchKernelGGL(topk2_seen_bits_kernel<96>, dim3((unsigned)((n_users + 63) / 64)), dim3(256), 0, s, seen_off,
                       seen_items, seen_rows, n_users, n_cand, (n_cand + 63) / 64, bits, only_if);
  CQL_LAUNCH_CHECK("topk2_seen_bits");
  return CQLREC_OK;
}

int cql_topk2_run(const QTk2Args& a, int d, hipStream_t s) {
  if (!cql_topk2_supported(d, a.k, a.n_cand)) return CQLREC_ERR_INVALID;
  if (cql_topk4_use(d, a.k, a.n_users, a.n_cand)) return cql_topk4_run(a, s);
  constexpr int smem = Tk2Cfg<128>::SMEM;
  static bool attr_set_dev[CQL_MAX_DEVICES] = {};   // > 64 KiB of dynamic LDS needs the opt-in once per kernel and device
  bool& attr_set = attr_set_dev[cql_device_slot()];
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)qtopk2_kernel<128, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)qtopk2_kernel<128, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  const int64_t rblks = (a.n_users + 255) / 256;
  static const bool force16 = getenv("CQL_TOPK4_KC") && atoi(getenv("CQL_TOPK4_KC")) == 16;      // A/B knob (both kernels)
  if (a.k <= 10 && !force16)
    hipLaunchKernelGGL((qtopk2_kernel<128, 10>), dim3((unsigned)(rblks * a.nsplit)), dim3(256), smem, s, a);
  else
    hipLaunchKernelGGL((qtopk2_kernel<128, 16>), dim3((unsigned)(rblks * a.nsplit)), dim3(256), smem, s, a);
  CQL_LAUNCH_CHECK("qtopk2");
  return CQLREC_OK;
}
