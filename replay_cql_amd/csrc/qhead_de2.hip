// Item-side backward of the Q-head, second form: ONE wave per SIMD, each wave owns TWO 32-item groups (64 items),
// software-pipelined inside the wave.  d = 128 (d = 256 keeps qde_kernel: the accumulators of two groups alone would need
// 256 registers; d = 64 is bound by the exponentials, not by LDS or MFMA, and keeps it too).
//
// Why (measured on qde_kernel, 32 items per wave, two waves per SIMD: 0.21 ms at cfg3, MFMA busy 49 %): every wave
// re-reads the whole state tile from LDS -- 8 row reads + 16 transposed reads + the strip = 20 KB -- for only 16 MFMAs.
// With all waves of a CU reading together that is 160 KB per 1024 MFMA cycles = 62 % of the LDS bandwidth, in bursts that
// the MFMA chains wait for: a timing-only build of that kernel WITHOUT any VALU work, LDS-DMA or barrier still needs
// 0.176 ms.  Here the same 20 KB feed 32 MFMAs (31 % of the LDS bandwidth), and the two independent item groups give
// the wave something to overlap:
//
//   slot 1   S0 = E0 . H^T   (8 MFMA)   |  16 transposed reads of this tile (for slots 3, 4)
//   slot 2   S1 = E1 . H^T   (8 MFMA)   |  P0 = exp2(S0 ...), column sums, bf16 pack            (57 VALU)
//   slot 3   dE0 += H^T P0   (8 MFMA)   |  P1                                                    (57 VALU)
//   slot 4   dE1 += H^T P1   (8 MFMA)   |  row reads + strip of the NEXT tile (12 ds_read_b128); at the end of a
//                                          stage: wait for the next stage's pieces, barrier, refill this stage's buffer
//
// The strip holds -lse in NATURAL units and is the C operand of the S chains (no initial-value registers); the item bias
// enters the exponent as a per-lane constant:  P = exp2((S - lse) log2e + b log2e).
// The accumulators of both groups (128 registers at d = 128) are touched by MFMAs only and live in the AGPR half of the
// file; everything the VALU reads stays below 256.
// Work decomposition, LDS image, staging and the deterministic two-piece sum of cut groups: as qde_kernel (qhead_de.hip).
#include <stdlib.h>
#include <type_traits>
#include "qhead_de_common.h"

#define QDE2_ITEMS 256      // items per group: 4 waves x 2 x 32
#ifndef QDE2_NBUF
#define QDE2_NBUF 2         // ring depth (stages of 64 states): with 2 the pieces a turn waits for were issued ONE stage
#endif                      // earlier and the wave parks on vmcnt(0) (PMC r02: 23 % of the wave's life in s_waitcnt / barrier)
#ifndef QDE2_VALU_PER_MFMA
#define QDE2_VALU_PER_MFMA 8   // VALU instructions the scheduler is asked to place behind each MFMA of slots 2 and 3
#endif

// out + scale * y as a product and a sum (never an fma): the terms and their order are those of the fix-up kernel and of
// the deferred fix-up (misc.hip is compiled without contraction), so every path to the gradient gives the same bits
__device__ __forceinline__ float qde2_axpy(float w, float scale, float u) {
#pragma clang fp contract(off)
  const float su = scale * u;
  return w + su;
}

template <int D, bool MASK>
__global__ __launch_bounds__(256, 1) void qde2_kernel(QDeArgs a) {
  using C = DeCfg<D, 4>;
  constexpr int KS = C::KS, FT = C::FT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // the ONLY LDS object of this kernel

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  const int64_t W = (int64_t)a.G * a.T;
  const int64_t u0 = (int64_t)blockIdx.x * W / gridDim.x, u1 = ((int64_t)blockIdx.x + 1) * W / gridDim.x;
  const int nst = (int)(u1 - u0);
  if (nst <= 0) return;
  unsigned long long stamp_tk = 0, stamp_rt = 0;
  if (a.stamps) qde_stamp(stamp_tk, stamp_rt);
  int g = (int)(u0 / a.T);
  int t = (int)(u0 - (int64_t)g * a.T);
  int t_seg = t;
  int t_dma = t;

  // ---- staging (see qde_kernel) ----------------------------------------------------------------------------------
  __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)a.H_b, 0, (int)(a.n_states * C::ROWB), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)a.nlse2, 0, (int)(a.n_states * 4), 0x00020000);
  uint32_t voff;
  {
    const int sub = lane >> 5, r7 = (lane >> 2) & 7, slot = lane & 3;
    const int rg0 = wave / C::PPG, hc = wave % C::PPG;
    const int q2 = (r7 >> 2) | ((rg0 & 1) << 1);             // 4 / PPG is even for d = 64, 128: parity independent of i
    voff = (uint32_t)((rg0 * 8 + r7) * C::ROWB + (8 * hc + 4 * sub + (slot ^ q2)) * 16);
  }
  const uint32_t voff_strip = (uint32_t)lane * 4;
  const uint32_t smem_base = lds_addr_of(smem);
  auto issue = [&](int stage_t, int buf) {
    const uint32_t bufp = __builtin_amdgcn_readfirstlane(smem_base + buf * C::BUF_BYTES);
    const uint32_t soff = (uint32_t)stage_t * C::STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < C::LPS; ++i) bdma16(voff, rs_h, soff + C::PSTEP * i, bufp + (4 * i + wave) * 1024);
    if (wave == (stage_t & 3)) bdma4(voff_strip, rs_s, (uint32_t)stage_t * (C::TI * 4), bufp + C::STAGE_BYTES);
  };

  // ---- read geometry (see qde_kernel).  Two sets of per-lane bases, one per ring buffer; they swap at the end of every
  // stage, so the stage body is written once and every LDS address in it is "base register + immediate"
  const lds_u8* lbase = (const lds_u8*)smem;
  const lds_u8 *pA0, *pA1, *pT0, *pT1, *pS;          // current buffer
  const lds_u8 *nA0, *nA1, *nT0, *nT1, *nS;          // the next buffer of the ring
  int oa0, oa1, ot0, ot1, os;
  {
    const int g1 = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
    oa0 = C::RG_BYTES * (r >> 3) + 64 * (r & 7) + 16 * ((0 + h) ^ ((r >> 2) & 3));
    oa1 = C::RG_BYTES * (r >> 3) + 64 * (r & 7) + 16 * ((2 + h) ^ ((r >> 2) & 3));
    ot0 = 64 * (4 * h + q) + 16 * ((2 * g1 + (p >> 1)) ^ ((0 + h) & 3)) + 8 * (p & 1);
    ot1 = 64 * (4 * h + q) + 16 * ((2 * g1 + (p >> 1)) ^ ((2 + h) & 3)) + 8 * (p & 1);
    os = C::STAGE_BYTES + 16 * h;
  }
  // per-lane bases of buffers `bc` (current) and `bn` (next): recomputed when the ring turns (10 adds per 64 MFMAs)
  auto set_ptrs = [&](int bc, int bn) {
    const lds_u8* c0 = lbase + bc * C::BUF_BYTES;
    const lds_u8* n0 = lbase + bn * C::BUF_BYTES;
    pA0 = c0 + oa0; pA1 = c0 + oa1; pT0 = c0 + ot0; pT1 = c0 + ot1; pS = c0 + os;
    nA0 = n0 + oa0; nA1 = n0 + oa1; nT0 = n0 + ot0; nT1 = n0 + ot1; nS = n0 + os;
  };
  set_ptrs(0, 1 % QDE2_NBUF);

  // ---- owner state: two 32-item groups per wave ----------------------------------------------------------------------
  bf16x8 rf[2][KS];
  float bl2[2];            // bias * log2e of this lane's item in each group
  f32x16 y[2][FT];
  float cs[2];
  auto load_owner = [&](int grp) {
    float bv[2];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      int64_t row = (int64_t)grp * QDE2_ITEMS + wave * 64 + gi * 32 + r;
      if (row >= a.n_items) row = a.n_items - 1;
#pragma unroll
      for (int s = 0; s < KS; ++s) rf[gi][s] = *reinterpret_cast<const bf16x8*>(a.E_b + row * D + 16 * s + 8 * h);
      bv[gi] = a.bias[row];
    }
    // These ordinary loads must be retired -- in hipcc's own bookkeeping too -- before the next LDS-DMA is issued: its
    // counted waits assume that nothing younger than its loads is in flight (cdna_hip_programming.md 5, trap (b)).  The
    // builtin is a wait the compiler models (vmcnt(0) only: 0x0F70); an empty asm with "+v" operands would do as well but
    // pins the fragments to the VGPR half, and the MFMA operands then get copied to AGPRs in every loop trip.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      bl2[gi] = bv[gi] * CQL_LOG2E;
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int i = 0; i < 16; ++i) y[gi][ft][i] = 0.f;
      cs[gi] = 0.f;
    }
  };
  auto store_piece = [&](int grp, bool first) {
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      const int64_t row = (int64_t)grp * QDE2_ITEMS + wave * 64 + gi * 32 + r;
      const bool ok = row < a.n_items;
      const float csum = cs[gi] + __shfl_xor(cs[gi], 32);
      if (first) {
        if (ok) {
          float* dst = a.out + row * D;
#pragma unroll
          for (int ft = 0; ft < FT; ++ft)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float4* pd = reinterpret_cast<float4*>(dst + ft * 32 + 8 * q + 4 * h);
              float4 o;
              if (a.accumulate) {
                const float4 old = *pd;
                o = make_float4(qde2_axpy(old.x, a.scale, y[gi][ft][4 * q + 0]), qde2_axpy(old.y, a.scale, y[gi][ft][4 * q + 1]),
                                qde2_axpy(old.z, a.scale, y[gi][ft][4 * q + 2]), qde2_axpy(old.w, a.scale, y[gi][ft][4 * q + 3]));
              } else {
                o = make_float4(a.scale * y[gi][ft][4 * q + 0], a.scale * y[gi][ft][4 * q + 1],
                                a.scale * y[gi][ft][4 * q + 2], a.scale * y[gi][ft][4 * q + 3]);
              }
              *pd = o;
            }
          if (h == 0) a.out_cs[row] = a.accumulate ? qde2_axpy(a.out_cs[row], a.scale, csum) : a.scale * csum;
        }
      } else {
        const int64_t srow = (int64_t)blockIdx.x * QDE2_ITEMS + wave * 64 + gi * 32 + r;
        float* dst = a.slab + srow * D;
#pragma unroll
        for (int ft = 0; ft < FT; ++ft)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(dst + ft * 32 + 8 * q + 4 * h) =
                make_float4(y[gi][ft][4 * q + 0], y[gi][ft][4 * q + 1], y[gi][ft][4 * q + 2], y[gi][ft][4 * q + 3]);
        if (h == 0) a.slab_cs[srow] = csum;
      }
    }
  };

  // ---- the software pipeline ---------------------------------------------------------------------------------------
  // Per tile (32 states x this wave's 2 x 32 items): four MFMA chains of 8 -- A = S0, B = S1 (scores of the two item
  // groups), C = dE0, D = dE1 (their gradient products) -- and 32 "half-chunks" of VALU work (P0 then P1: 2 fma + exp /
  // exp + 2 add + pack for two probabilities).  Measured on this chip (tools/probes/mfma_probe.hip, one wave per SIMD): an
  // MFMA followed by a half-chunk issues every 34 cycles, a whole 7-instruction chunk behind one MFMA makes the gap 46
  // cycles, one ds_read_b128 per gap adds ~5.  So ONE half-chunk per MFMA, all 32 gaps of a tile used, and the chains of
  // THREE tiles in flight -- the "period" of tile t:
  //
  //   gap     MFMA                          VALU              LDS (one read per gap, into the OTHER register set)
  //   0-4     D(t-1), last 5                P0(t)  0-4        gap 4: [ring turn if t ends a stage]
  //   5-12    B(t)   (rows of t, read in    P0(t)  5-12       gaps 4-15: strip + rows of tile t+1
  //                   period t-1)
  //   13-20   C(t)                          P0 13-15, P1 0-4
  //   21-28   A(t+1)                        P1(t)  5-12       gaps 16-31: transposed reads of tile t+1
  //   29-31   D(t), first 3                 P1(t) 13-15
  //
  // A(t) was done in period t-1; P0 reads its accumulator from gap 0 on (the chain ended 3 gaps earlier), P1 reads B's
  // from gap 16 (3 gaps behind it); C's first four products need words 0-3 of P0 (gap 7), the others words 4-7 (gap 15);
  // D's first four need words 0-3 of P1 (gap 23), its others words 4-7 (gap 31) -- they run in gaps 0-4 of the next period.
  // A group (or the block's range) ends: the period runs without A(t+1), the 5 products of D drain behind it.
#define QDE2_FENCE() __builtin_amdgcn_sched_barrier(0)
  bf16x8 af[2][KS];         // row fragments, by tile parity inside the stage (those of tile t+1 are read while B(t) and
  f32x16 sv[2];             // C(t) still run on tile t's); -lse of the tile's states = C operand of its S chains
  f32x16 acc0, acc1;        // S accumulators: A writes acc0, B writes acc1
  bf16x8 tf[2][FT][2];      // transposed fragments, by tile parity inside the stage
  bf16x8 dpa, dpb;          // P1 fragments of the previous tile (for the D products still pending)
  float ht0 = 0.f, ht1 = 0.f;      // the two exponent arguments / values travelling from half A to half B of a chunk
  // half-chunks of the exponentials.  Volatile asm: hipcc would otherwise regroup them (sinks the column sums to the end
  // of the stage as v_pk_add_f32 and keeps every exponential alive).  Half B reads a v_exp result one instruction later
  // at the earliest (gfx950: one wait state behind a transcendental).
  auto half_a = [&](const f32x16& acc, int k, float b0, float b1) {
#ifdef QDE2_ABL_NOCHUNK      // timing-only build: no exponentials
    ht0 = acc[2 * k]; ht1 = acc[2 * k + 1];
    return;
#endif
    asm volatile(
        "v_fmamk_f32 %0, %2, 0x3fb8aa3b, %4\n\t"
        "v_fmamk_f32 %1, %3, 0x3fb8aa3b, %5\n\t"
        "v_exp_f32 %0, %0"
        : "=&v"(ht0), "=&v"(ht1)
        : "v"(acc[2 * k]), "v"(acc[2 * k + 1]), "v"(b0), "v"(b1));
  };
  auto half_b = [&](uint32_t& w, float& csum) {
#ifdef QDE2_ABL_NOCHUNK
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=&v"(w) : "v"(ht0), "v"(ht1));
    return;
#endif
    asm volatile(
        "v_exp_f32 %1, %1\n\t"
        "v_add_f32 %3, %3, %0\n\t"
        "v_add_f32 %3, %3, %1\n\t"
        "v_cvt_pk_bf16_f32 %2, %0, %1"
        : "+v"(ht0), "+v"(ht1), "=&v"(w), "+v"(csum));
  };
  auto frag = [](const uint32_t (&pw)[8], int s2) {
    u32x4 v = {pw[4 * s2 + 0], pw[4 * s2 + 1], pw[4 * s2 + 2], pw[4 * s2 + 3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  // LDS reads of the tile FOLLOWING tile (cur buffer, IT): NIT = its index in its stage, from the other buffer if IT is
  // the last tile of the stage.  idx 0..3 strip, 4..11 rows, 12..27 transposed.
  auto next_read = [&](auto IT, int idx) {
#if defined(QDE2_ABL_NOROWS) && defined(QDE2_ABL_NOTR)      // timing-only build: no LDS reads in the loop
    if (t >= 0) return;
#elif defined(QDE2_ABL_NOROWS)
    if (idx < 4 + KS) return;
#elif defined(QDE2_ABL_NOTR)
    if (idx >= 4 + KS) return;
#endif
    constexpr bool END = decltype(IT)::value == C::TILES - 1;
    constexpr int NIT = END ? 0 : decltype(IT)::value + 1;
    constexpr int noff = NIT * C::TILE_BYTES;
    if (idx < 4) {
      const f32x4 t4 = *(const lds_f4*)((END ? nS : pS) + 128 * NIT + 32 * idx);
      sv[NIT & 1][4 * idx + 0] = t4[0];
      sv[NIT & 1][4 * idx + 1] = t4[1];
      sv[NIT & 1][4 * idx + 2] = t4[2];
      sv[NIT & 1][4 * idx + 3] = t4[3];
    } else if (idx < 4 + KS) {
      const int sa = idx - 4;
      af[NIT & 1][sa] = *(const lds_bf16x8*)(((sa & 1) ? (END ? nA1 : pA1) : (END ? nA0 : pA0)) + noff + 512 * (sa >> 1));
    } else {
      const int q = idx - 4 - KS, ft = q >> 2, s2 = (q >> 1) & 1, jj = q & 1;
      const bf16x4 t4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
          (lds_bf16x4*)((jj ? (END ? nT1 : pT1) : (END ? nT0 : pT0)) + noff + C::RG_BYTES * (2 * s2 + jj) + 512 * ft));
      tf[NIT & 1][ft][s2][4 * jj + 0] = t4[0];
      tf[NIT & 1][ft][s2][4 * jj + 1] = t4[1];
      tf[NIT & 1][ft][s2][4 * jj + 2] = t4[2];
      tf[NIT & 1][ft][s2][4 * jj + 3] = t4[3];
    }
  };
  // the ring turns (tile = last of its stage): the next stage's pieces have landed for everyone, everyone has left this
  // stage's buffer (its last reads were issued in the previous period), which is refilled with stage + 2
  // `younger` = stages issued behind the one the turn waits for (0 .. QDE2_NBUF - 2): their pieces -- LPS per stage, plus
  // the strip of the stages whose strip THIS wave loads -- may stay in flight (vmcnt is counted in issue order)
#ifdef QDE2_STAMP_TURN      // diagnostic build: where a ring turn parks the wave (ticks of wave 0 per block, three intervals)
  unsigned long long tt_lgkm = 0, tt_vm = 0, tt_bar = 0, tt_cal = 0;
  {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    tt_cal = c1 - c0;            // what two stamps with nothing in between read
  }
#define QDE2_TSTAMP(x) const unsigned long long x = __builtin_amdgcn_s_memtime()
#else
#define QDE2_TSTAMP(x)
#endif
  auto ring_turn = [&](bool more, bool refill, int cur_buf, int younger) {
    if (more) {
      QDE2_TSTAMP(ts0);
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's reads of the buffer refilled below (issued >= 4 gaps ago)
      QDE2_TSTAMP(ts1);
      if constexpr (QDE2_NBUF == 3) {
        // at most ONE younger stage: its LPS row pieces, plus the strip if this wave loaded it -- three literals, two
        // scalar branches (a general count through a switch costs a chain of ~40 scalar branches per stage: measured,
        // 8 % of the kernel)
        if (younger <= 0 || a.T < QDE2_NBUF) {
          de_wait_vmcnt<0>();
        } else {
          const int tq = (t_dma == 0) ? a.T - 1 : t_dma - 1;       // in-group index of the youngest stage issued
          if (wave == (tq & 3)) de_wait_vmcnt<C::LPS + 1>();
          else de_wait_vmcnt<C::LPS>();
        }
      } else {
        de_wait_vmcnt<0>();     // (QDE2_NBUF == 2: nothing younger; deeper rings: conservative)
      }
      QDE2_TSTAMP(ts2);
#ifndef QDE_ABL_NOBAR
      __builtin_amdgcn_s_barrier();
#endif
      QDE2_TSTAMP(ts3);
#ifdef QDE2_STAMP_TURN
      tt_lgkm += ts1 - ts0; tt_vm += ts2 - ts1; tt_bar += ts3 - ts2;
#endif
#ifdef QDE_ABL_NODMA
      if (false) {
#else
      if (refill) {
#endif
        issue(t_dma, cur_buf);
        if (++t_dma == a.T) t_dma = 0;
      }
    }
  };
  // A chain of the tile whose rows are in af / sv, on its own (first tile of a group)
  auto chain_a_alone = [&]() {      // (a piece starts with the first tile of a stage: parity 0)
    acc0 = sv[0];
#pragma unroll
    for (int s = 0; s < KS; ++s) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][s], rf[0][s], acc0, 0, 0, 0);
  };
  // the last 5 products of D of tile (parity IT), on their own (last tile of a group)
  auto drain_d = [&](auto IT) {
    constexpr int P = decltype(IT)::value & 1;
#pragma unroll
    for (int m = 3; m < 2 * FT; ++m) {
      const int ft = m % FT, s2 = m / FT;
      y[1][ft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[P][ft][s2], s2 ? dpb : dpa, y[1][ft], 0, 0, 0);
    }
  };
  // one period.  There is ONE form per tile parity: at the first tile of a piece the pending D products run on zeroed
  // P fragments (they add nothing), at its last tile the A chain of the following tile is computed in vain (the next
  // piece redoes it with its own owner fragments) -- run-time variants of the period would duplicate the chains under a
  // branch, and hipcc then shuffles the accumulators at every join.
  auto period = [&](auto IT, bool more, bool refill, int cur_buf, int younger) {
    constexpr int P = decltype(IT)::value & 1;                 // register set of this tile's transposed fragments
    constexpr bool END = decltype(IT)::value == C::TILES - 1;
    const int64_t left = MASK ? (a.n_states - ((int64_t)t * C::TI + 32 * decltype(IT)::value)) : 32;
    uint32_t pw0[8], pw1[8];
    float c0 = 0.f, c1 = 0.f;
    bf16x8 pa0 = {}, pb0 = {}, pa1 = {};
    auto bias_of = [&](int gi, int elem) {       // exponent addend of accumulator element `elem`; -inf past the batch's end
      if constexpr (MASK) return (mfma_row(elem, h) < left) ? bl2[gi] : NEG_INF_F;
      else return bl2[gi];
    };
#pragma unroll
    for (int gp = 0; gp < 32; ++gp) {
      // ---- MFMA of this gap
      if (gp < 5) {
        const int m = 3 + gp, ft = m % FT, s2 = m / FT;
        y[1][ft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[P ^ 1][ft][s2], s2 ? dpb : dpa, y[1][ft], 0, 0, 0);
      } else if (gp < 13) {
        const int s = gp - 5;
        if (s == 0) acc1 = sv[P];
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[P][s], rf[1][s], acc1, 0, 0, 0);
      } else if (gp < 21) {
        const int m = gp - 13, ft = m % FT, s2 = m / FT;
        if (m == 0) pa0 = frag(pw0, 0);
        if (m == FT) pb0 = frag(pw0, 1);
        y[0][ft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[P][ft][s2], s2 ? pb0 : pa0, y[0][ft], 0, 0, 0);
      } else if (gp < 29) {
        const int s = gp - 21;
        if (s == 0) acc0 = sv[P ^ 1];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[P ^ 1][s], rf[0][s], acc0, 0, 0, 0);
      } else {
        const int m = gp - 29, ft = m % FT;
        if (m == 0) pa1 = frag(pw1, 0);
        y[1][ft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[P][ft][0], pa1, y[1][ft], 0, 0, 0);
      }
      QDE2_FENCE();
      // ---- half-chunk of this gap: P0 in gaps 0-15 (from acc0), P1 in gaps 16-31 (from acc1)
      {
        const int hc = gp & 15, k = hc >> 1;
        if (gp < 16) {
          if ((hc & 1) == 0) half_a(acc0, k, bias_of(0, 2 * k), bias_of(0, 2 * k + 1));
          else half_b(pw0[k], c0);
        } else {
          if ((hc & 1) == 0) half_a(acc1, k, bias_of(1, 2 * k), bias_of(1, 2 * k + 1));
          else half_b(pw1[k], c1);
        }
      }
      QDE2_FENCE();
      // ---- LDS reads of the next tile, one per gap: strip + rows in gaps 4-15 (second register set: B(t) and A(t+1)
      // run on different sets), transposed fragments in gaps 16-31.  The ring turns in front of the first of them.
      if (gp == 4) {
        if constexpr (END) ring_turn(more, refill, cur_buf, younger);
      }
      if (gp >= 4) next_read(IT, gp - 4);
      QDE2_FENCE();
    }
    cs[0] += c0;
    cs[1] += c1;
    dpa = pa1;
    dpb = frag(pw1, 1);
  };

  load_owner(g);
#pragma unroll
  for (int pp = 0; pp < 2; ++pp)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) tf[pp][ft][s2] = bf16x8{};     // finite: the first tile's pending-D products add 0 * this

  // ---- prologue: stages 0 and 1 in flight; rows, strip and transposed fragments of the first tile in registers ----
  int issued = 0;
  for (int s0 = 0; s0 < QDE2_NBUF && s0 < nst; ++s0) {
    issue(t_dma, s0);
    ++issued;
    if (++t_dma == a.T) t_dma = 0;
  }
  de_wait_vmcnt<0>();        // (once per launch: the whole ring, not only stage 0)
  __builtin_amdgcn_s_barrier();
  {   // "the tile following tile -1": read through next_read with buffer 0 in the role of the NEXT buffer
    set_ptrs(1 % QDE2_NBUF, 0);
#pragma unroll
    for (int idx = 0; idx < 4 + KS + 4 * FT; ++idx) next_read(std::integral_constant<int, C::TILES - 1>{}, idx);
    set_ptrs(0, 1 % QDE2_NBUF);
  }

  // pieces of item groups (outer) x stages of the piece (inner): the owner fragments are invariant in the inner loop
  int j = 0, cur_buf = 0;
  while (j < nst) {
    int seg_end = j + (a.T - t);
    if (seg_end > nst) seg_end = nst;
    chain_a_alone();                      // A of the piece's first tile (its rows are in registers)
    dpa = bf16x8{};                       // nothing pending from a previous tile of this group
    dpb = bf16x8{};
    for (; j < seg_end; ++j) {
      const bool more = j + 1 < nst, refill = issued < nst;
      static_assert(C::TILES == 2, "two tiles per stage");
      const int younger = issued - (j + 2);        // stages in flight behind stage j + 1 when the ring turns
      period(std::integral_constant<int, 0>{}, true, false, cur_buf, 0);
      period(std::integral_constant<int, 1>{}, more, refill, cur_buf, younger > 0 ? younger : 0);
      if (more && refill) ++issued;
      ++t;
      // the ring turned: the next buffer is the current one now
      cur_buf = (cur_buf + 1 == QDE2_NBUF) ? 0 : cur_buf + 1;
      set_ptrs(cur_buf, (cur_buf + 1 == QDE2_NBUF) ? 0 : cur_buf + 1);
    }
    drain_d(std::integral_constant<int, 1>{});
    store_piece(g, t_seg == 0);
    if (j < nst) {
      ++g;
      t = 0;
      t_seg = 0;
      load_owner(g);
    }
  }
  if (a.stamps) {
    unsigned long long tk, rt;
    qde_stamp(tk, rt);
    if (tid == 0) {
      a.stamps[2 * blockIdx.x] = tk - stamp_tk;
#ifdef QDE2_STAMP_TURN
      (void)rt;
      a.stamps[2 * blockIdx.x + 1] = (tt_cal << 60) | ((tt_lgkm & 0xFFFFF) << 40) | ((tt_vm & 0xFFFFF) << 20) | (tt_bar & 0xFFFFF);
#else
      a.stamps[2 * blockIdx.x + 1] = rt - stamp_rt;
#endif
    }
  }
}

// =============================================================================================================
// host side
// =============================================================================================================
template <int D, bool MASK>
static void qde2_launch_n(const QDeArgs& a, int grid, hipStream_t s) {
  constexpr int smem = QDE2_NBUF * DeCfg<D, 4>::BUF_BYTES;
  if constexpr (smem > 64 * 1024) {     // > 64 KiB of dynamic LDS needs the opt-in once per kernel and device
    static bool attr_set_dev[CQL_MAX_DEVICES] = {};
    bool& attr_set = attr_set_dev[cql_device_slot()];
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)qde2_kernel<D, MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      attr_set = true;
    }
  }
  hipLaunchKernelGGL((qde2_kernel<D, MASK>), dim3(grid), dim3(256), smem, s, a);
}

// rows [0, n_items): `a` prepared by cql_qde_launch (G, T for 256-item groups; nlse2 = -lse in NATURAL units here)
int cql_qde2_run(const QDeArgs& a, int d, int grid, hipStream_t s) {
  // whole stages only (the caller keeps the generic form for batches that are not a multiple of 64 states)
  if (d != 128 || (a.n_states % DeCfg<128, 4>::TI) != 0) return CQLREC_ERR_INVALID;
  qde2_launch_n<128, false>(a, grid, s);
  return CQLREC_OK;
}
