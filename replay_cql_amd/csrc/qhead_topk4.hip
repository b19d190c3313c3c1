// Top-K scoring pass, third form (d = 128, many users): qtopk2_kernel (qhead_topk2.hip) with FOUR 32-user groups per wave
// instead of two -- 512 users per block, so that every 16 KiB item stage that is staged into LDS (LDS-DMA: ~60 cycles
// of the issuing wave per KiB) and every fragment that is read from it feeds twice as many MFMAs.  Measured on
// qtopk2_kernel (profiles/r03_topk_ablation.txt): of its 1.75 ms per 65 536 users x 100 000 items the LDS-DMA refills
// are 0.30, the LDS reads 0.10 and the stage barriers 0.05 -- all three per STAGE, i.e. halved here per MFMA.
//
// What the reference does here: CQL._predict scores every (user, item) pair, drops seen items
// (replay/models/base_rec.py:417-464 _filter_seen), keeps the k largest per user (replay/utils.py:59-127).
//
// Same contract, same selection logic, same output format as qtopk2_kernel (keys = order-preserving score bits << 32 |
// ~row; per (slice, user, lane half) a sorted list of 16; topk_merge_kernel merges them).  What had to change:
//  * the four groups' user fragments (128 registers) live in AccVGPRs and enter the score products as the B operand of
//    inline-asm MFMAs (the lists of the four groups take the other 128 AccVGPRs): the VGPR half holds four score
//    accumulators, two sets of item-row fragments and bias, and the epilogue's scratch;
//  * the per-lane candidate queues shrink from 24 to 12 entries (16 queues of 6 KiB per block instead of 8 of 12 KiB): the
//    slow path walks the quads that hold a candidate in a RUN-TIME loop with one merge site in it (the queue is merged as
//    soon as a lane holds more than 8 keys, a quad adds at most 4), instead of one straight-line scan behind one merge site;
//  * the seen words of a wave's 128 users are 1 KiB of mask words in LDS per stage: streamed in as one LDS-DMA piece
//    (two 64-user blocks of the bitmap), or -- LISTS, the default -- expanded by the wave from a 256-byte list of the
//    stage's seen (user, item) cells (tk4_lists_kernel, at the end of this file), which spares the 1.6 GB bitmap.
#include <stdlib.h>
#include <type_traits>
#include "qhead_de_common.h"

#define TK4_G 4          // user groups per wave
#define TK4_Q 12         // queue entries per lane and group (16 queues of 6 KiB per block beside the 61 KiB ring)
#define TK4_K 16
#define TK4_NBUF 3
#define TK4_SLOT_WORDS 64
#define TK4_SLOT_CAP 124       // 256 bytes = 2 header words + 124 16-bit entries (lanes 2..63 of the wave hold two each)
static_assert(TK4_K == QS_TOPK_K, "list format shared with topk_merge_kernel");

typedef __attribute__((address_space(3))) unsigned long long lds_u64_4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_4;
typedef __attribute__((address_space(3))) u32x2_4 lds_u2_4;
typedef __attribute__((address_space(3))) unsigned int lds_u32_4;
typedef __attribute__((address_space(3))) u32x4 lds_u4_4;

template <int D>
struct Tk4Cfg {
  using C = DeCfg<D, 4>;
  static constexpr int STRIP = 256;                      // ONE bias strip (64 floats) per stage, loaded by wave (stage & 3)
  static constexpr int SEEN = 4 * 1024;                  // per wave: 128 users x 2 words
  static constexpr int BUF = C::STAGE_BYTES + STRIP + SEEN;
  static constexpr int RING = TK4_NBUF * BUF;
  static constexpr int QUEUES = 4 * TK4_G * TK4_Q * 512;
  static constexpr int LISTS = TK4_NBUF * 4 * 256;      // LISTS form: 256 B (128 entries) per wave and ring buffer
  static constexpr int SMEM = RING + QUEUES + LISTS;
};

#define TK4_AW(dst, src) asm("v_accvgpr_write_b32 %0, %1" : "=a"(dst) : "v"(src))
#define TK4_AR(dst, src) asm("v_accvgpr_read_b32 %0, %1" : "=v"(dst) : "a"(src))

__device__ __forceinline__ float tk4_max4(float a, float b, float c, float d) {
  float t, q;
  asm("v_max_f32 %0, %1, %2" : "=v"(t) : "v"(c), "v"(d));
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(q) : "v"(a), "v"(b), "v"(t));
  return q;
}
__device__ __forceinline__ float tk4_max3(float a, float b, float c) {
  float q;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(q) : "v"(a), "v"(b), "v"(c));
  return q;
}
// score products: A = item-row fragment (VGPRs), B = user fragment (AccVGPRs), C/D = score accumulator (VGPRs).
// First product of a chain: C = the bias strip registers; the others accumulate in place.
__device__ __forceinline__ void tk4_mfma_first(f32x16& acc, const bf16x8& a_frag, const u32x4& b_acc, const f32x16& c) {
  const u32x4 av = __builtin_bit_cast(u32x4, a_frag);
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(av), "a"(b_acc), "v"(c));
}
__device__ __forceinline__ void tk4_mfma_acc(f32x16& acc, const bf16x8& a_frag, const u32x4& b_acc) {
  const u32x4 av = __builtin_bit_cast(u32x4, a_frag);
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "a"(b_acc));
}

// LISTS: the seen filter arrives as entry lists (tk4_lists_* below: one 256-byte slot per wave of 128 users and stage of
// 64 items, longer lists continue in an overflow area) instead of the dense bitmap; the wave expands the list of a
// stage into the same 1 KiB of mask words in LDS that the dense form streams in.
// KC: list entries that are kept sorted (10 for k <= 10, else 16): an insertion costs ~9 instructions per entry it
// passes, and what falls off position KC - 1 of a (slice, user, lane half) list cannot be among that list's best k.
template <int D, bool LISTS, int KC>
__global__ __launch_bounds__(256, 1) void qtopk4_kernel(QTk2Args a) {
  static_assert(KC == 10 || KC == 16, "lists of 10 or 16 sorted keys");
  // the two forms are launched one behind the other; the word the list builder leaves picks one on the device
  if (a.guard != nullptr && ((*a.guard != 0u) != (a.guard_want != 0))) return;
  using T = Tk4Cfg<D>;
  using C = typename T::C;
  constexpr int KS = C::KS;
  static_assert(D == 128 && C::TILES == 2 && C::LPS == 4, "d = 128: two tiles per stage, four row pieces per wave");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // the ONLY LDS object of this kernel

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int split = blockIdx.x % a.nsplit;
  const int64_t rblk = blockIdx.x / a.nsplit;
  const int64_t res0 = rblk * 512 + wave * 128;
  const int64_t s_begin = (int64_t)split * a.split_rows;
  const int64_t s_end = (s_begin + a.split_rows < a.n_cand) ? (s_begin + a.split_rows) : a.n_cand;
  const int nstage = (s_end > s_begin) ? (int)((s_end - s_begin + C::TI - 1) / C::TI) : 0;
  const uint32_t gstage0 = (uint32_t)(s_begin / C::TI);

  // ---- user fragments -> AccVGPRs --------------------------------------------------------------------------------------
  u32x4 rfa[TK4_G][KS];
#pragma unroll
  for (int g = 0; g < TK4_G; ++g) {
    int64_t row = res0 + g * 32 + r;
    if (row >= a.n_users) row = a.n_users - 1;
#pragma unroll
    for (int s = 0; s < KS; ++s) rfa[g][s] = *reinterpret_cast<const u32x4*>(a.H_b + row * D + 16 * s + 8 * h);   // only ever an
                                                                                // "a" operand: hipcc keeps it in AccVGPRs
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);       // (see qde2_kernel::load_owner)
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- staging ---------------------------------------------------------------------------------------------------------
  __amdgpu_buffer_rsrc_t rs_e = __builtin_amdgcn_make_buffer_rsrc((void*)a.E_b, 0, (int)(a.n_cand * C::ROWB), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, (int)(a.n_cand * 4), 0x00020000);
  // seen words of this wave's 128 users = two consecutive 64-user blocks of the bitmap ([block][stage][64 users][2 words]):
  // lanes 0..31 fetch the first block's 512 B of a stage, lanes 32..63 the second block's.  A block past n_users does not
  // exist in the bitmap: its lanes get an offset past num_records (they read 0).  No filter = an empty buffer.
  // LISTS: this wave's row of 256-byte slots ([wave of 128 users][stage]: count, overflow offset, 124 entries).
  const int64_t nst_all = (a.n_cand + C::TI - 1) / C::TI;
  const int64_t nblk64 = LISTS ? (a.n_users + 127) / 128 : (a.n_users + 63) / 64;
  const int64_t blk0 = LISTS ? (res0 >> 7) : (res0 >> 6);
  const uint32_t* const seen_src = LISTS ? a.seen_lists : a.seen_bits;
  const bool has_seen = seen_src != nullptr && blk0 < nblk64;          // wave-uniform
  const uint32_t* wsrc = has_seen ? seen_src + blk0 * nst_all * (LISTS ? 64 : 128) : (const uint32_t*)a.bias;
  const bool two_blocks = !LISTS && has_seen && (blk0 + 1 < nblk64);
  const int64_t w_records = has_seen ? (LISTS ? nst_all * 256 : (two_blocks ? 2 : 1) * nst_all * 512) : 0;   // < 2^31: host
  // (wave-uniform by construction; made so for the compiler too, which otherwise hands the asm a VGPR tuple)
  const uint64_t wp = (uint64_t)wsrc;
  const uint64_t wp_u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(wp >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wp);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)wp_u, 0, __builtin_amdgcn_readfirstlane((int)w_records),
                                                                  0x00020000);
  const uint32_t voff_w = LISTS ? (uint32_t)lane * 4u
                                : (uint32_t)((lane >> 5) * (uint32_t)(nst_all * 512) + (lane & 31) * 16);
  uint32_t voff;
  {
    const int sub = lane >> 5, r7 = (lane >> 2) & 7, slot = lane & 3;
    const int rg0 = wave / C::PPG, hc = wave % C::PPG;
    const int q2 = (r7 >> 2) | ((rg0 & 1) << 1);
    voff = (uint32_t)((rg0 * 8 + r7) * C::ROWB + (8 * hc + 4 * sub + (slot ^ q2)) * 16);
  }
  const uint32_t voff4 = (uint32_t)lane * 4;
  const uint32_t smem_base = lds_addr_of(smem);
  // piece pc of stage `stage` into ring buffer `buf`: 0..3 rows, 4 seen words, 5 the bias strip (one wave per stage)
  auto issue_piece = [&](int stage, int buf, int pc) __attribute__((always_inline)) {
    const uint32_t bufp = __builtin_amdgcn_readfirstlane(smem_base + buf * T::BUF);
    const uint32_t gs = gstage0 + (uint32_t)stage;
    if (pc < C::LPS) bdma16(voff, rs_e, gs * C::STAGE_BYTES + C::PSTEP * pc, bufp + (4 * pc + wave) * 1024);
    else if (pc == C::LPS) {
      if constexpr (LISTS)
        bdma4(voff_w, rs_w, gs * 256, __builtin_amdgcn_readfirstlane(smem_base + T::RING + T::QUEUES + (buf * 4 + wave) * 256));
      else bdma16(voff_w, rs_w, gs * 512, bufp + C::STAGE_BYTES + T::STRIP + wave * 1024);
    }
    else if (wave == (stage & 3)) bdma4(voff4, rs_b, gs * (C::TI * 4), bufp + C::STAGE_BYTES);
  };

  // ---- read geometry ------------------------------------------------------------------------------------------------------
  const lds_u8* lbase = (const lds_u8*)smem;
  const int oa0 = C::RG_BYTES * (r >> 3) + 64 * (r & 7) + 16 * ((0 + h) ^ ((r >> 2) & 3));
  const int oa1 = C::RG_BYTES * (r >> 3) + 64 * (r & 7) + 16 * ((2 + h) ^ ((r >> 2) & 3));
  const int os = C::STAGE_BYTES + 16 * h;
  const int ow = C::STAGE_BYTES + T::STRIP + wave * 1024 + r * 8;      // + 256 g for user group g
  const lds_u8 *pA0, *pA1, *pS, *pW;
  const lds_u8 *nA0, *nA1, *nS;
  int boff_c = 0, boff_n = T::BUF;
  auto set_ptrs = [&]() __attribute__((always_inline)) {
    pA0 = lbase + boff_c + oa0; pA1 = lbase + boff_c + oa1; pS = lbase + boff_c + os; pW = lbase + boff_c + ow;
    nA0 = lbase + boff_n + oa0; nA1 = lbase + boff_n + oa1; nS = lbase + boff_n + os;
  };
  set_ptrs();

  // ---- per-lane selection state -----------------------------------------------------------------------------------------
  uint32_t la[TK4_G][2 * TK4_K];      // sorted lists (AccVGPRs): key j of group g = la[g][2j] | la[g][2j+1] << 32
  float thr[TK4_G];
  int cnt[TK4_G];
  lds_u8* const qb0 = (lds_u8*)smem + T::RING + ((wave * TK4_G) * TK4_Q * 64 + lane) * 8;     // group g: + g * TK4_Q * 512
#pragma unroll
  for (int g = 0; g < TK4_G; ++g) {
#ifdef TK4_ABL_NOSLOW       // timing-only build: nothing ever beats the bound
    thr[g] = 3.0e38f;
#else
    thr[g] = -3.0e38f;      // finite: an item that scores -inf is never selected
#endif
    cnt[g] = 0;
#pragma unroll
    for (int j = 0; j < 2 * TK4_K; ++j) TK4_AW(la[g][j], 0u);
  }
  const int kth = a.k - 1;
  const bool kb0 = kth & 1, kb1 = kth & 2, kb2 = kth & 4, kb3 = kth & 8;

  // the queue of group g goes into its list; new bound (qtopk2_kernel::merge)
  auto merge = [&](auto G_) __attribute__((always_inline)) {
    constexpr int g = decltype(G_)::value;
    lds_u8* const qb = qb0 + g * (TK4_Q * 512);
    unsigned long long lst[TK4_K];
#pragma unroll
    for (int j = 0; j < TK4_K; ++j) {
      if (j < KC) {
        uint32_t lo, hi;
        TK4_AR(lo, la[g][2 * j]);
        TK4_AR(hi, la[g][2 * j + 1]);
        lst[j] = ((unsigned long long)hi << 32) | lo;
      } else {
        lst[j] = 0ull;         // (never holds a key: the registers behind it stay zero from the start)
      }
    }
#pragma unroll 1
    for (int e = 0; __builtin_amdgcn_ballot_w64(e < cnt[g]) != 0; ++e) {
      unsigned long long kx = (e < cnt[g]) ? *(const lds_u64_4*)(qb + 512 * e) : 0ull;
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
      TK4_AW(la[g][2 * j], (uint32_t)lst[j]);
      TK4_AW(la[g][2 * j + 1], (uint32_t)(lst[j] >> 32));
    }
    cnt[g] = 0;
    unsigned long long kk;
    if (kth == KC - 1) {       // (wave-uniform) the usual case, k = the number of keys kept sorted: the last of them
      kk = lst[KC - 1];
    } else {                   // any other k: a 4-level select tree over the bits of k - 1 (~90 instructions)
      const unsigned long long m0 = kb0 ? ~0ull : 0ull, m1 = kb1 ? ~0ull : 0ull, m2 = kb2 ? ~0ull : 0ull,
                               m3 = kb3 ? ~0ull : 0ull;
      unsigned long long t8[8], t4[4], t2[2];
#pragma unroll
      for (int j = 0; j < 8; ++j) t8[j] = (lst[2 * j + 1] & m0) | (lst[2 * j] & ~m0);
#pragma unroll
      for (int j = 0; j < 4; ++j) t4[j] = (t8[2 * j + 1] & m1) | (t8[2 * j] & ~m1);
#pragma unroll
      for (int j = 0; j < 2; ++j) t2[j] = (t4[2 * j + 1] & m2) | (t4[2 * j] & ~m2);
      kk = (t2[1] & m3) | (t2[0] & ~m3);
    }
    const uint32_t okey = (uint32_t)(kk >> 32);
    const float own = (kk != 0ull) ? f32_from_order_key(okey) : -3.0e38f;
    const float own_up = (kk != 0ull) ? f32_from_order_key(okey + 1u) : -3.0e38f;
    thr[g] = fmaxf(own_up, __shfl_xor(own, 32));
  };

  // slow path of the epilogue of one (tile, group): the tile's four quads in a run-time loop, ONE merge site
  auto slow = [&](const f32x16& acc, auto G_, int64_t trow0, uint32_t w) __attribute__((always_inline)) {
    constexpr int g = decltype(G_)::value;
    lds_u8* const qb = qb0 + g * (TK4_Q * 512);
    const uint32_t wh = w >> (4 * h);
    const int lim = (int)((s_end - trow0 < 64) ? (s_end - trow0) : 64) - 4 * h;   // element admissible iff rc < lim
    const uint32_t nrow = ~((uint32_t)trow0 + 4u * (uint32_t)h);                  // ~(row0 + rc) = nrow - rc
    // which quads hold a candidate at all: four quad maxima (straight-line), one wave-uniform bit each
    uint32_t qmask = 0u;
    {
      const float m0 = tk4_max4(acc[0], acc[1], acc[2], acc[3]), m1 = tk4_max4(acc[4], acc[5], acc[6], acc[7]);
      const float m2 = tk4_max4(acc[8], acc[9], acc[10], acc[11]), m3 = tk4_max4(acc[12], acc[13], acc[14], acc[15]);
      qmask = (__builtin_amdgcn_ballot_w64(m0 >= thr[g]) != 0 ? 1u : 0u) | (__builtin_amdgcn_ballot_w64(m1 >= thr[g]) != 0 ? 2u : 0u) |
              (__builtin_amdgcn_ballot_w64(m2 >= thr[g]) != 0 ? 4u : 0u) | (__builtin_amdgcn_ballot_w64(m3 >= thr[g]) != 0 ? 8u : 0u);
    }
#pragma unroll 1
    while (qmask != 0u) {
      const int q = __builtin_ctz(qmask);          // wave-uniform
      qmask &= qmask - 1u;
      // the quad's four scores by bit-selects under two wave-uniform masks (3 v_bfi per element; `q1 ? a : b` on a
      // run-time q became s_set_gpr_idx register indexing: 7 instructions per element)
      float e[4];
#ifdef TK4_ABL_QUAD_SELECT      // A/B build: the selects as written first
      const bool q1 = q & 1, q2 = q & 2;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float lo = q1 ? acc[4 + jj] : acc[jj];
        const float hi = q1 ? acc[12 + jj] : acc[8 + jj];
        e[jj] = q2 ? hi : lo;
      }
#else
      const uint32_t m1 = (q & 1) ? 0xFFFFFFFFu : 0u, m2 = (q & 2) ? 0xFFFFFFFFu : 0u;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const uint32_t lo = (m1 & __float_as_uint(acc[4 + jj])) | (~m1 & __float_as_uint(acc[jj]));
        const uint32_t hi = (m1 & __float_as_uint(acc[12 + jj])) | (~m1 & __float_as_uint(acc[8 + jj]));
        e[jj] = __uint_as_float((m2 & hi) | (~m2 & lo));
      }
#endif
      if (__builtin_amdgcn_ballot_w64(cnt[g] > TK4_Q - 4) != 0) merge(G_);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int rc = 8 * q + jj;
        const bool c0 = e[jj] >= thr[g];
        if (c0 && (((wh >> rc) & 1u) == 0u) && (rc < lim)) {
          const unsigned long long key =
              ((unsigned long long)f32_order_key(e[jj]) << 32) | (unsigned long long)(nrow - (uint32_t)rc);
          *(lds_u64_4*)(qb + 512 * cnt[g]) = key;
          cnt[g] += 1;
        }
      }
    }
  };

#define TK4_FENCE() __builtin_amdgcn_sched_barrier(0)
  bf16x8 af[2][KS];        // item-row fragments and bias by tile parity
  f32x16 sv[2];
  f32x16 acc0, acc1, acc2, acc3;      // (four names, not an array: hipcc kept one element of an array of them in scratch)
#pragma unroll
  for (int i = 0; i < 16; ++i) acc3[i] = NEG_INF_F;       // "the tile before the first": nothing beats any bound
  u32x2_4 wv[TK4_G];       // seen words of the stage per group: .x tile 0, .y tile 1
#pragma unroll
  for (int g = 0; g < TK4_G; ++g) wv[g] = u32x2_4{0u, 0u};
  uint32_t w_pend = 0u;    // seen word of the previous stage's last tile for the LAST group (its epilogue is still pending)
  float qm[4], tmx = 0.f;

  auto next_read = [&](auto IT, int idx) __attribute__((always_inline)) {      // rows 0..7, bias 8..11 of the following tile
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
  // epilogue pieces of a finished chain, spread over gaps 3..6 of the chain that follows it
  auto epi = [&](int gp, const f32x16& ac, auto G_, int64_t trow0, uint32_t w) __attribute__((always_inline)) {
    constexpr int g = decltype(G_)::value;
    if (gp == 3) {
      qm[0] = tk4_max3(ac[0], ac[1], ac[2]);
      qm[1] = tk4_max3(ac[3], ac[4], ac[5]);
      qm[2] = tk4_max3(ac[6], ac[7], ac[8]);
    } else if (gp == 4) {
      qm[3] = tk4_max3(ac[9], ac[10], ac[11]);
      tmx = tk4_max3(ac[12], ac[13], ac[14]);
    } else if (gp == 5) {
      qm[0] = tk4_max3(qm[0], qm[1], qm[2]);
      tmx = tk4_max3(qm[3], tmx, ac[15]);
      asm("v_max_f32 %0, %1, %2" : "=v"(tmx) : "v"(qm[0]), "v"(tmx));
    } else if (gp == 6) {
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(tmx >= thr[g]) != 0, 0)) slow(ac, G_, trow0, w);
    }
  };

  int st = 0, buf_c = 0;
  bool refill = false;
  auto ring_turn = [&]() __attribute__((always_inline)) {
    if (st + 1 < nstage) {
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
      // the stage now being entered has landed once at most the younger stages' pieces are outstanding (5 or 6 per stage:
      // the wave that loaded a younger stage's bias strip over-waits by one piece)
      const int last = (nstage - 1 < st + TK4_NBUF - 1) ? nstage - 1 : st + TK4_NBUF - 1;   // youngest stage issued
      if (last - (st + 1) <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    refill = st + TK4_NBUF < nstage;
  };

  // LISTS: the stage's slot (this wave's 256 staging bytes: word 0 = entries of the list, word 1 = where its part beyond
  // TK4_SLOT_CAP continues in the overflow area, then the 16-bit entries (user in wave) << 6 | item in stage) becomes the
  // stage's mask words -- clear them, set the bits -- in the gaps of the stage's first chain, one small step per gap
  // (the wave's LDS operations execute in order: no wait between the steps beyond the one for the entries).
  uint32_t ls_n = 0u, ls_e2 = 0u;
  auto lists_step = [&](int gp) __attribute__((always_inline)) {
    lds_u8* const mW = (lds_u8*)smem + boff_c + C::STAGE_BYTES + T::STRIP + wave * 1024;
    const lds_u8* const stg = (const lds_u8*)smem + T::RING + T::QUEUES + (buf_c * 4 + wave) * 256;
    auto set_bit = [&](uint32_t code) __attribute__((always_inline)) {      // word (user, tile) = code >> 5, bit = code & 31
      __hip_atomic_fetch_or((lds_u32_4*)(mW + ((code >> 5) << 2)), 1u << (code & 31u), __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_WAVEFRONT);
    };
    if (gp == 0) {
      ls_n = has_seen ? *(const lds_u32_4*)(stg) : 0u;
      ls_e2 = *(const lds_u32_4*)(stg + 4 * lane);
    } else if (gp == 1) {
      *(lds_u4_4*)(mW + 16 * lane) = u32x4{0u, 0u, 0u, 0u};
      ls_n = ls_n < 8192u ? ls_n : 8192u;     // (128 users x 64 items: no list is longer; bounds the loop below whatever was read)
    } else if (gp == 2 || gp == 3) {
      const uint32_t n_slot = ls_n < (uint32_t)TK4_SLOT_CAP ? ls_n : (uint32_t)TK4_SLOT_CAP;
      const uint32_t idx = 2u * (uint32_t)lane - 4u + (uint32_t)(gp - 2);     // lanes 0, 1 hold the header: idx wraps, never < n_slot
      if (idx < n_slot) set_bit((ls_e2 >> (16 * (gp - 2))) & 0xFFFFu);
    } else if (gp == 4) {
      if (__builtin_expect(ls_n > (uint32_t)TK4_SLOT_CAP, 0)) {      // a stage of popular items: the rest, from global memory
        const uint32_t ovf0 = *(const lds_u32_4*)(stg + 4);
        const uint32_t n_ovf = __builtin_amdgcn_readfirstlane((int)(ls_n - (uint32_t)TK4_SLOT_CAP));
        for (uint32_t i = (uint32_t)lane; i < n_ovf; i += 64u) set_bit((uint32_t)a.seen_lists_ovf[(uint64_t)ovf0 + i]);
      }
    }
  };

  using G0 = std::integral_constant<int, 0>;
  using G1 = std::integral_constant<int, 1>;
  using G2 = std::integral_constant<int, 2>;
  using G3 = std::integral_constant<int, 3>;
  // one tile: chains of groups 0..3; behind chain g the epilogue of the chain before it (group 3's: the previous tile's)
  auto period = [&](auto IT, int64_t row0, int64_t row0_prev) __attribute__((always_inline)) {
    constexpr int P = decltype(IT)::value;
    constexpr bool END = P == 1;
    const uint32_t w3_prev = END ? wv[3][0] : w_pend;            // seen word of the previous tile for group 3
    // ---- chain 0 | epilogue of group 3 (previous tile) | rows 0..7 of the next tile
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (END && s == 0) ring_turn();      // in front of the first read of the next stage's buffer
      if (s == 0) tk4_mfma_first(acc0, af[P][0], rfa[0][0], sv[P]);
      else tk4_mfma_acc(acc0, af[P][s], rfa[0][s]);
      TK4_FENCE();
      epi(s, acc3, G3{}, row0_prev, w3_prev);
      TK4_FENCE();
      next_read(IT, s);
      TK4_FENCE();
      if constexpr (LISTS && !END) {
        if (s < 5) lists_step(s);
        TK4_FENCE();
      }
    }
    if constexpr (!END) {      // the new stage's seen words (group 3's pending one of the old stage has just been used)
#pragma unroll
      for (int g = 0; g < TK4_G; ++g) wv[g] = *(const lds_u2_4*)(pW + 256 * g);
    }
    // ---- chain 1 | epilogue of group 0 | bias of the next tile
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s == 0) tk4_mfma_first(acc1, af[P][0], rfa[1][0], sv[P]);
      else tk4_mfma_acc(acc1, af[P][s], rfa[1][s]);
      TK4_FENCE();
      epi(s, acc0, G0{}, row0, END ? wv[0][1] : wv[0][0]);
      TK4_FENCE();
      if (s < 4) next_read(IT, KS + s);
      TK4_FENCE();
    }
    // ---- chain 2 | epilogue of group 1 | the refill of the buffer just left, one piece per gap
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s == 0) tk4_mfma_first(acc2, af[P][0], rfa[2][0], sv[P]);
      else tk4_mfma_acc(acc2, af[P][s], rfa[2][s]);
      TK4_FENCE();
#ifndef TK4_ABL_NODMA
      if (END && s < 6 && refill) issue_piece(st + TK4_NBUF, buf_c, s);
#endif
      TK4_FENCE();
      epi(s, acc1, G1{}, row0, END ? wv[1][1] : wv[1][0]);
      TK4_FENCE();
    }
    // ---- chain 3 | epilogue of group 2
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s == 0) tk4_mfma_first(acc3, af[P][0], rfa[3][0], sv[P]);
      else tk4_mfma_acc(acc3, af[P][s], rfa[3][s]);
      TK4_FENCE();
      epi(s, acc2, G2{}, row0, END ? wv[2][1] : wv[2][0]);
      TK4_FENCE();
    }
  };

  if (nstage > 0) {
    // ---- prologue: the whole ring in flight; rows and bias of the first tile in registers -------------------------------
    for (int s0 = 0; s0 < TK4_NBUF && s0 < nstage; ++s0)
#pragma unroll
      for (int pc = 0; pc < 6; ++pc) issue_piece(s0, s0, pc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {   // "the tile following the last tile of the buffer before buffer 0": next_read with the buffers' roles swapped
      const lds_u8 *kA0 = nA0, *kA1 = nA1, *kS = nS;
      nA0 = pA0; nA1 = pA1; nS = pS;
#pragma unroll
      for (int idx = 0; idx < KS + 4; ++idx) next_read(std::integral_constant<int, 1>{}, idx);
      nA0 = kA0; nA1 = kA1; nS = kS;
    }
    int64_t row_prev = s_begin;
    for (st = 0; st < nstage; ++st) {
      const int64_t row0 = s_begin + (int64_t)st * C::TI;
      period(std::integral_constant<int, 0>{}, row0, row_prev);
      period(std::integral_constant<int, 1>{}, row0 + 32, row0);
      row_prev = row0 + 32;
      w_pend = wv[3][1];
      boff_c = boff_n;
      boff_n = (boff_n + T::BUF == T::RING) ? 0 : boff_n + T::BUF;
      buf_c = (buf_c + 1 == TK4_NBUF) ? 0 : buf_c + 1;
      set_ptrs();
    }
    // the last chain's epilogue
#pragma unroll
    for (int gp = 3; gp <= 6; ++gp) epi(gp, acc3, G3{}, row_prev, w_pend);
  }

  // ---- flush the queues, write the lists ---------------------------------------------------------------------------------
  merge(G0{});
  merge(G1{});
  merge(G2{});
  merge(G3{});
#pragma unroll
  for (int g = 0; g < TK4_G; ++g) {
    const int64_t row = res0 + g * 32 + r;
    if (row < a.n_users) {
      unsigned long long* dst = a.keys + (((int64_t)split * a.n_users + row) * 2 + h) * TK4_K;
#pragma unroll
      for (int j = 0; j < TK4_K; j += 2) {
        uint32_t w0, w1, w2, w3;
        TK4_AR(w0, la[g][2 * j]);
        TK4_AR(w1, la[g][2 * j + 1]);
        TK4_AR(w2, la[g][2 * j + 2]);
        TK4_AR(w3, la[g][2 * j + 3]);
        *reinterpret_cast<uint4*>(dst + j) = make_uint4(w0, w1, w2, w3);
      }
    }
  }
}

// ---- the seen filter as entry lists ------------------------------------------------------------------------------------
// A dense bitmap of 131 072 users x 100 000 items is 1.64 GB written per launch of the scoring kernel (0.25-0.45 ms at the
// HBM write rate) to say ~9 M things.  Here: one 256-byte slot per (wave of 128 users, stage of 64 items) -- word 0 the
// number of entries, word 1 where the list continues, then up to 124 16-bit entries (user in wave) << 6 | item in stage
// -- and an overflow area for the longer lists (the stages of the most popular items: on a Zipf log nearly every user
// has seen them).  Built in the bitmap's own space by six small launches + one scan; if the overflow area is too small
// (users who have seen a large part of the catalogue) the word behind the space is set, the bitmap is built over it
// after all and the scoring kernel of that form runs: both forms are always enqueued, the word picks on the device.
struct Tk4Lists {
  uint32_t* slots;      // [rows][stages][64]
  uint16_t* ovf;        // [rows][row_cap]
  int64_t rows, row_cap;
};
static Tk4Lists tk4_lists_carve(void* base, int64_t space_bytes, int64_t n_users, int64_t n_cand) {
  Tk4Lists L;
  L.rows = (n_users + 127) / 128;
  const int64_t slot_bytes = L.rows * ((n_cand + 63) / 64) * (TK4_SLOT_WORDS * 4);
  L.slots = (uint32_t*)base;
  L.ovf = (uint16_t*)((char*)base + slot_bytes);
  int64_t cap = (space_bytes - slot_bytes) / 2 / (L.rows > 0 ? L.rows : 1);
  if (cap * L.rows > 0xFFFFFFFFll) cap = 0xFFFFFFFFll / L.rows;       // an offset is a 32-bit word of the slot header
  L.row_cap = cap > 0 ? cap / 8 * 8 : 0;
  return L;
}
// One block per row (128 users), modelled on topk2_seen_bits_kernel: the users' seen lists are ascending, so two
// threads per user walk them once per chunk of TK4_LCH stages; the chunk's slots are assembled in LDS (counts by LDS
// atomics, then the entries) and leave with one coalesced write each -- only the 64-byte parts a list reaches.  The part
// of a list beyond TK4_SLOT_CAP goes to the row's overflow area, in stage order (header word 1 = where).
#define TK4_LCH 96
__global__ __launch_bounds__(256) void tk4_lists_kernel(const int64_t* __restrict__ seen_off,
                                                        const int32_t* __restrict__ seen_items,
                                                        const int32_t* __restrict__ seen_rows, int64_t n_users,
                                                        int64_t n_cand, int64_t nst_all, uint32_t* __restrict__ slots,
                                                        uint16_t* __restrict__ ovf, int64_t row_cap, uint32_t* __restrict__ flag) {
  __shared__ __attribute__((aligned(16))) uint32_t tile[TK4_LCH * TK4_SLOT_WORDS];     // the chunk's slots
  __shared__ uint32_t fillpos[TK4_LCH], ovbase[TK4_LCH];
  __shared__ uint32_t s_run, s_lost, s_w0;
  const int t = threadIdx.x, user = t >> 1, q = t & 1;
  const int64_t u = (int64_t)blockIdx.x * 128 + user;
  int64_t cur = 0, end = 0;                 // this user's entries not yet placed (kept in step by its two threads)
  if (u < n_users) {
    const int64_t srow = seen_rows ? (int64_t)seen_rows[u] : u;
    cur = seen_off[srow];
    end = seen_off[srow + 1];
  }
  const int64_t first = cur;                // (an id repeated in the ascending list counts once: a list never exceeds 128 x 64)
  if (t == 0) { s_run = 0u; s_lost = 0u; }
  uint32_t* const dst = slots + (int64_t)blockIdx.x * nst_all * TK4_SLOT_WORDS;
  const uint64_t row_base = (uint64_t)blockIdx.x * (uint64_t)row_cap;
  const uint32_t ucode = (uint32_t)user << 6;
  for (int64_t st0 = 0; st0 < nst_all; st0 += TK4_LCH) {
    const int nst = (int)((nst_all - st0 < TK4_LCH) ? (nst_all - st0) : TK4_LCH);
    for (int i = t; i < nst; i += 256) { tile[i * TK4_SLOT_WORDS] = 0u; fillpos[i] = 0u; }
    __syncthreads();
    const int64_t item_lo = st0 * 64, item_hi = (st0 + nst) * 64;
    // ---- count
    int64_t j = cur + q;
    for (; j < end; j += 2) {
      const int32_t id = seen_items[j];
      if (id >= item_hi) break;
      if (id >= item_lo && id < n_cand && !(j > first && seen_items[j - 1] == id))
        atomicAdd(&tile[(int)((id >> 6) - st0) * TK4_SLOT_WORDS], 1u);
    }
    __syncthreads();
    // ---- where the long lists continue: exclusive prefix of the excess over the chunk's stages (two waves, shuffles)
    static_assert(TK4_LCH <= 128, "the prefix below uses two waves");
    {
      const int lane_ = t & 63;
      uint32_t ex = 0u;
      if (t < nst) {
        const uint32_t n = tile[t * TK4_SLOT_WORDS];
        ex = n > TK4_SLOT_CAP ? n - TK4_SLOT_CAP : 0u;
      }
      uint32_t inc = ex;
      if (t < 128) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t v = __shfl_up(inc, o);
          if (lane_ >= o) inc += v;
        }
        if (t == 63) s_w0 = inc;          // total of the first wave
      }
      __syncthreads();
      if (t < nst) {
        const uint32_t excl = s_run + inc - ex + (t >= 64 ? s_w0 : 0u);
        ovbase[t] = excl;
        tile[t * TK4_SLOT_WORDS + 1] = (uint32_t)(row_base + excl);
      }
      __syncthreads();
      if (t == 127) s_run += inc + s_w0;  // (lane 63 of the second wave: its inclusive sum + the first wave's total)
    }
    __syncthreads();
    // ---- place the entries (the same walk again)
    int64_t j2 = cur + q;
    for (; j2 < end; j2 += 2) {
      const int32_t id = seen_items[j2];
      if (id >= item_hi) break;
      if (id >= item_lo && id < n_cand && !(j2 > first && seen_items[j2 - 1] == id)) {
        const int sl = (int)((id >> 6) - st0);
        const uint32_t pos = atomicAdd(&fillpos[sl], 1u);
        const uint16_t code = (uint16_t)(ucode | (uint32_t)(id & 63));
        if (pos < TK4_SLOT_CAP) {
          reinterpret_cast<uint16_t*>(&tile[sl * TK4_SLOT_WORDS])[4 + pos] = code;
        } else {
          const uint64_t o = (uint64_t)ovbase[sl] + (pos - TK4_SLOT_CAP);
          if (o < (uint64_t)row_cap) ovf[row_base + o] = code;
          else s_lost = 1u;
        }
      }
    }
    if (j2 > end) j2 = end;
    {   // first entry at or beyond item_hi = the smaller of the two stopping points (ascending list, stride 2)
      const int64_t o1 = __shfl_xor(j2, 1, 2);
      cur = (o1 < j2) ? o1 : j2;
    }
    __syncthreads();
    // ---- write the slots out: 16 words (64 bytes) at a time, as far as the list reaches
    uint4* out = reinterpret_cast<uint4*>(dst + st0 * TK4_SLOT_WORDS);
    for (int i = t; i < nst * 16; i += 256) {
      const int sl = i >> 4, part = (i >> 2) & 3;          // 4 uint4 = one 64-byte part
      uint32_t n = tile[sl * TK4_SLOT_WORDS];
      n = n < TK4_SLOT_CAP ? n : TK4_SLOT_CAP;
      const int parts = (int)((8u + 2u * n + 63u) / 64u);   // header + entries, in 64-byte parts (>= 1)
      if (part < parts) out[i] = reinterpret_cast<const uint4*>(tile)[i];
    }
    __syncthreads();
  }
  if (t == 0 && s_lost != 0u) atomicOr(flag, 1u);
}

// =============================================================================================================
// host side
// =============================================================================================================
// The four-group form pays when the users alone (nearly) fill the chip with 512-user blocks.  With fewer users the
// catalogue is cut into slices to fill it, and every slice selects its own top k from scratch -- the selection work per
// wave grows with the number of (group, slice) pairs, not with the items scanned (k ln(N/k) insertions either way) --
// which is where 65 536 users in two slices lose what the MFMA schedule gained (1.65 against 1.60 ms for qtopk2_kernel;
// 131 072 users in one slice: 2.80 against 3.18 ms).  profiles/r03_topk_ablation.txt.
bool cql_topk4_use(int d, int k, int64_t n_users, int64_t n_cand) {
  static const int mode = getenv("CQL_TOPK4") ? atoi(getenv("CQL_TOPK4")) : 1;      // 0: never; 2: whenever the shape allows
  if (mode == 0 || d != 128 || k > TK4_K || n_cand * 256 >= (1ll << 31)) return false;
  if (((n_cand + 63) / 64) * 1024 >= (1ll << 31)) return false;                    // the seen descriptor of a wave
  return mode == 2 || n_users >= 512 * 160;
}

bool cql_topk4_lists_on() {
  static const bool on = !(getenv("CQL_TOPK4_LISTS") && getenv("CQL_TOPK4_LISTS")[0] == '0');     // A/B knob
  return on;
}

// does a launch of this shape have room for its lists in the bitmap's space?
bool cql_topk4_lists_fit(int64_t n_users, int64_t n_cand, int64_t space_bytes) {
  return tk4_lists_carve(nullptr, space_bytes, n_users, n_cand).row_cap >= 4096;
}

// entry lists of the users' seen items into `space` (the bitmap's space, space_bytes of it); *flag (outside that space)
// = 0: the scoring kernel takes the lists; 1: they did not fit, the caller's bitmap builder runs (under the same word)
int cql_topk4_seen_lists(const int64_t* seen_off, const int32_t* seen_items, const int32_t* seen_rows, int64_t n_users,
                         int64_t n_cand, void* space, int64_t space_bytes, uint32_t* flag, hipStream_t s) {
  const Tk4Lists L = tk4_lists_carve(space, space_bytes, n_users, n_cand);
  CQL_REQUIRE(L.row_cap >= 4096, "topk4_seen_lists: no room for the lists");
  if (hipMemsetAsync(flag, 0, 4, s) != hipSuccess) {
    cql_set_error("topk4_seen_lists: hipMemsetAsync failed");
    return CQLREC_ERR_HIP;
  }
  hipLaunchKernelGGL(tk4_lists_kernel, dim3((unsigned)L.rows), dim3(256), 0, s, seen_off, seen_items, seen_rows, n_users, n_cand,
                     (n_cand + 63) / 64, L.slots, L.ovf, L.row_cap, flag);
  CQL_LAUNCH_CHECK("topk4_seen_lists");
  return CQLREC_OK;
}

// a.seen_lists set (with a.guard = the builder's word): both forms are enqueued, the word picks one
int cql_topk4_run(const QTk2Args& a, hipStream_t s) {
  constexpr int smem = Tk4Cfg<128>::SMEM;
  static bool attr_set_dev[CQL_MAX_DEVICES] = {};   // > 64 KiB of dynamic LDS needs the opt-in once per kernel and device
  bool& attr_set = attr_set_dev[cql_device_slot()];
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)qtopk4_kernel<128, false, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)qtopk4_kernel<128, true, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)qtopk4_kernel<128, false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)qtopk4_kernel<128, true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  const int64_t rblks = (a.n_users + 511) / 512;
  const dim3 grid((unsigned)(rblks * a.nsplit));
  static const bool force16 = getenv("CQL_TOPK4_KC") && atoi(getenv("CQL_TOPK4_KC")) == 16;      // A/B knob
  const bool kc12 = a.k <= 10 && !force16;      // KC = 10
  auto launch = [&](const QTk2Args& b, bool lists) {
    if (lists) {
      if (kc12) hipLaunchKernelGGL((qtopk4_kernel<128, true, 10>), grid, dim3(256), smem, s, b);
      else hipLaunchKernelGGL((qtopk4_kernel<128, true, 16>), grid, dim3(256), smem, s, b);
    } else {
      if (kc12) hipLaunchKernelGGL((qtopk4_kernel<128, false, 10>), grid, dim3(256), smem, s, b);
      else hipLaunchKernelGGL((qtopk4_kernel<128, false, 16>), grid, dim3(256), smem, s, b);
    }
  };
  if (a.seen_lists != nullptr && a.guard != nullptr) {
    QTk2Args b = a;
    // the bitmap form FIRST: nearly always it returns at once, and in front of the real launch its blocks (each a whole
    // CU's LDS and registers) are handed out to an idle chip in ~4 us; behind it they waited for the real blocks to
    // leave (35 us on average, on the critical path of every launch)
    b.guard_want = 1;
    b.seen_lists = nullptr;
    launch(b, false);
    b.seen_lists = a.seen_lists;
    b.seen_lists_ovf = tk4_lists_carve((void*)a.seen_lists, 1ll << 40, a.n_users, a.n_cand).ovf;
    b.guard_want = 0;
    launch(b, true);
  } else {
    QTk2Args b = a;
    b.guard = nullptr;
    b.seen_lists = nullptr;
    launch(b, false);
  }
  CQL_LAUNCH_CHECK("qtopk4");
  return CQLREC_OK;
}
