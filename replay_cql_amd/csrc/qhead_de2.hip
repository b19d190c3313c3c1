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
#ifndef QDE2_VALU_PER_MFMA
#define QDE2_VALU_PER_MFMA 8   // VALU instructions the scheduler is asked to place behind each MFMA of slots 2 and 3
#endif

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
  const lds_u8 *nA0, *nA1, *nT0, *nT1, *nS;          // the other buffer
  {
    const int g1 = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
    const int oa0 = C::RG_BYTES * (r >> 3) + 64 * (r & 7) + 16 * ((0 + h) ^ ((r >> 2) & 3));
    const int oa1 = C::RG_BYTES * (r >> 3) + 64 * (r & 7) + 16 * ((2 + h) ^ ((r >> 2) & 3));
    const int ot0 = 64 * (4 * h + q) + 16 * ((2 * g1 + (p >> 1)) ^ ((0 + h) & 3)) + 8 * (p & 1);
    const int ot1 = 64 * (4 * h + q) + 16 * ((2 * g1 + (p >> 1)) ^ ((2 + h) & 3)) + 8 * (p & 1);
    const int os = C::STAGE_BYTES + 16 * h;
    pA0 = lbase + oa0; pA1 = lbase + oa1; pT0 = lbase + ot0; pT1 = lbase + ot1; pS = lbase + os;
    nA0 = pA0 + C::BUF_BYTES; nA1 = pA1 + C::BUF_BYTES; nT0 = pT0 + C::BUF_BYTES; nT1 = pT1 + C::BUF_BYTES;
    nS = pS + C::BUF_BYTES;
  }

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
              float4 o = make_float4(a.scale * y[gi][ft][4 * q + 0], a.scale * y[gi][ft][4 * q + 1],
                                     a.scale * y[gi][ft][4 * q + 2], a.scale * y[gi][ft][4 * q + 3]);
              if (a.accumulate) {
                const float4 old = *pd;
                o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
              }
              *pd = o;
            }
          if (h == 0) a.out_cs[row] = a.accumulate ? a.out_cs[row] + a.scale * csum : a.scale * csum;
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

  // ---- the pieces of a tile -----------------------------------------------------------------------------------------------
  bf16x8 af[KS];      // row fragments of the tile whose S chains run next (read one slot 4 ahead)
  f32x16 sv;          // -lse of that tile's states = C operand of its S chains
  auto read_rows = [&](auto NEXT, auto IT) {        // 8 + 4 ds_read_b128; NEXT: from the other ring buffer
    constexpr int toff = decltype(IT)::value * C::TILE_BYTES;
    const lds_u8* b0 = decltype(NEXT)::value ? nA0 : pA0;
    const lds_u8* b1 = decltype(NEXT)::value ? nA1 : pA1;
    const lds_u8* bs = decltype(NEXT)::value ? nS : pS;
#pragma unroll
    for (int s = 0; s < KS; ++s) af[s] = *(const lds_bf16x8*)(((s & 1) ? b1 : b0) + toff + 512 * (s >> 1));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 t4 = *(const lds_f4*)(bs + 128 * decltype(IT)::value + 32 * q);
      sv[4 * q + 0] = t4[0];
      sv[4 * q + 1] = t4[1];
      sv[4 * q + 2] = t4[2];
      sv[4 * q + 3] = t4[3];
    }
  };
  // One tile = four slots of 8 MFMAs, written in the order they are meant to ISSUE: every MFMA is followed by its share of
  // the other pipes' work, and __builtin_amdgcn_sched_barrier(0) after each piece keeps hipcc's scheduler from regrouping
  // them (left alone it issues the chains back to back and the exponentials in one lump behind them).
  //   exponentials of a group, in 8 chunks of two elements: 2 fma, 2 exp2, 2 adds (column sum), 1 bf16 pack = word k of
  //   the B operand of the second chain (k-order = accumulator row order).
#define QDE2_FENCE() __builtin_amdgcn_sched_barrier(0)
  // exponentials of a group in 8 chunks of two elements: 2 fma, 2 exp2, 2 adds (column sum), 1 bf16 pack = word k of the
  // B operand of the second chain (k-order = accumulator row order).  ONE volatile asm statement per chunk: hipcc otherwise
  // sinks the adds to the end of the stage (and packs them into v_pk_add_f32), keeping all 32 exponentials of a tile alive,
  // and lowers the pack of two separately converted values to 4 instructions.  Hazards inside: the two v_exp results are read
  // one instruction later at the earliest (gfx950: one wait state behind a transcendental); the accumulator registers read
  // here were written by an MFMA chain that ended at least three MFMAs earlier (slot layout below).
  auto chunk = [&](const f32x16& acc, int gi, int k, uint32_t (&pw)[8], float& csum, int64_t left) {
    float t0, t1;
    uint32_t w;
#if defined(QDE2_ABL_NOCHUNK)      // timing-only build: one pack instead of the 7-instruction chunk
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=&v"(w) : "v"(acc[2 * k]), "v"(acc[2 * k + 1]));
    t0 = t1 = 0.f;
    (void)csum; (void)left;
    pw[k] = w;
    return;
#elif defined(QDE2_ABL_NOEXP)      // timing-only build: the transcendentals become moves
    asm volatile(
        "v_fmamk_f32 %0, %4, 0x3fb8aa3b, %6\n\t"
        "v_fmamk_f32 %1, %5, 0x3fb8aa3b, %6\n\t"
        "v_mov_b32 %0, %0\n\t"
        "v_mov_b32 %1, %1\n\t"
        "v_add_f32 %3, %3, %0\n\t"
        "v_add_f32 %3, %3, %1\n\t"
        "v_cvt_pk_bf16_f32 %2, %0, %1"
        : "=&v"(t0), "=&v"(t1), "=&v"(w), "+v"(csum)
        : "v"(acc[2 * k]), "v"(acc[2 * k + 1]), "v"(bl2[gi]));
    (void)left;
    pw[k] = w;
    return;
#endif
    if constexpr (!MASK) {
      asm volatile(
          "v_fmamk_f32 %0, %4, 0x3fb8aa3b, %6\n\t"
          "v_fmamk_f32 %1, %5, 0x3fb8aa3b, %6\n\t"
          "v_exp_f32 %0, %0\n\t"
          "v_exp_f32 %1, %1\n\t"
          "v_add_f32 %3, %3, %0\n\t"
          "v_add_f32 %3, %3, %1\n\t"
          "v_cvt_pk_bf16_f32 %2, %0, %1"
          : "=&v"(t0), "=&v"(t1), "=&v"(w), "+v"(csum)
          : "v"(acc[2 * k]), "v"(acc[2 * k + 1]), "v"(bl2[gi]));
    } else {      // states past the end of the batch contribute nothing: exp2(-inf) = 0
      const float m0 = (mfma_row(2 * k, h) < left) ? bl2[gi] : NEG_INF_F;
      const float m1 = (mfma_row(2 * k + 1, h) < left) ? bl2[gi] : NEG_INF_F;
      asm volatile(
          "v_fmamk_f32 %0, %4, 0x3fb8aa3b, %6\n\t"
          "v_fmamk_f32 %1, %5, 0x3fb8aa3b, %7\n\t"
          "v_exp_f32 %0, %0\n\t"
          "v_exp_f32 %1, %1\n\t"
          "v_add_f32 %3, %3, %0\n\t"
          "v_add_f32 %3, %3, %1\n\t"
          "v_cvt_pk_bf16_f32 %2, %0, %1"
          : "=&v"(t0), "=&v"(t1), "=&v"(w), "+v"(csum)
          : "v"(acc[2 * k]), "v"(acc[2 * k + 1]), "v"(m0), "v"(m1));
    }
    pw[k] = w;
  };
  auto frag = [](const uint32_t (&pw)[8], int s2) {
    u32x4 v = {pw[4 * s2 + 0], pw[4 * s2 + 1], pw[4 * s2 + 2], pw[4 * s2 + 3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  // One tile = four slots of 8 MFMAs (d = 128), written in the order they are meant to ISSUE; a fence after each piece keeps
  // hipcc's scheduler from regrouping them (left alone it issues the chains back to back and the exponentials in one lump).
  //   slot 1   S0 chain, 2 transposed reads of this tile behind each MFMA
  //   slot 2   S1 chain; from its 4th MFMA on one chunk of P0 behind each MFMA (5 chunks)
  //   slot 3   dE0 chain (the s2 = 0 products first: they need words 0..3); chunks 5..7 of P0 behind its first three
  //            MFMAs, chunks 0..4 of P1 behind the others
  //   slot 4   dE1 chain; chunks 5..7 of P1 behind its first three MFMAs; [end of a stage: the ring turns]; the 12 row
  //            reads of the next tile behind the others
  // `more`: another stage follows in this block's range; END: last tile of its stage (`cur_buf` = the ring buffer this
  // stage lives in, refilled with stage + 2).
  static_assert(KS == 8, "slot layout written for d = 128");
  constexpr int LAG = 3;      // MFMAs between the end of an S chain and the first read of its accumulator
  auto tile = [&](auto IT, auto END, bool more, bool refill, int cur_buf) {
    constexpr int toff = decltype(IT)::value * C::TILE_BYTES;
    const int64_t left = MASK ? (a.n_states - ((int64_t)t * C::TI + 32 * decltype(IT)::value)) : 32;
    f32x16 acc0 = sv, acc1 = sv;
    bf16x8 tf[FT][2];
    uint32_t pw0[8], pw1[8];
    float c0 = 0.f, c1 = 0.f;
    // ---- slot 1
    constexpr int TRP = (4 * FT) / KS;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], rf[0][s], acc0, 0, 0, 0);
      QDE2_FENCE();
#pragma unroll
      for (int e = 0; e < TRP; ++e) {
        const int idx = s * TRP + e, ft = idx >> 2, s2 = (idx >> 1) & 1, jj = idx & 1;
#ifdef QDE2_ABL_NOTR       // timing-only build: no transposed reads
        bf16x4 t4 = {af[s][0], af[s][1], af[s][2], af[s][3]};
        asm volatile("" : "+v"(t4));
#else
        const bf16x4 t4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (lds_bf16x4*)((jj ? pT1 : pT0) + toff + C::RG_BYTES * (2 * s2 + jj) + 512 * ft));
#endif
        tf[ft][s2][4 * jj + 0] = t4[0];
        tf[ft][s2][4 * jj + 1] = t4[1];
        tf[ft][s2][4 * jj + 2] = t4[2];
        tf[ft][s2][4 * jj + 3] = t4[3];
      }
      QDE2_FENCE();
    }
    // ---- slot 2 (every index below is a compile-time constant after unrolling: a counter carried through the loops
    // would leave the arrays runtime-indexed when SROA runs, i.e. in scratch memory)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], rf[1][s], acc1, 0, 0, 0);
      QDE2_FENCE();
      if (s >= LAG) {
        chunk(acc0, 0, s - LAG, pw0, c0, left);                 // chunks 0 .. 4
        QDE2_FENCE();
      }
    }
    // ---- the ring turns here at the end of a stage: the transposed reads of this tile (slot 1) are back by now, the row
    // reads of the next tile -- from the other buffer -- start right behind it, a slot and a half before their S chains
    // (the reads are issued unconditionally: behind the block's last stage they fetch stale LDS bytes nobody uses -- a
    // run-time branch around them would duplicate the MFMA chains and make hipcc shuffle the accumulators at the join)
    constexpr int NIT = decltype(END)::value ? 0 : 1;
    constexpr int noff = NIT * C::TILE_BYTES;
    const lds_u8* b0 = decltype(END)::value ? nA0 : pA0;
    const lds_u8* b1 = decltype(END)::value ? nA1 : pA1;
    const lds_u8* bs = decltype(END)::value ? nS : pS;
    if constexpr (decltype(END)::value) {
      if (more) {
        __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0): nobody still reads the buffer that is refilled below
        de_wait_vmcnt<0>();                       // this wave's pieces of the next stage have landed
#ifndef QDE_ABL_NOBAR
        __builtin_amdgcn_s_barrier();             // everyone's have; everyone left this stage's buffer
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
    }
    // read `idx` of the next tile: 0..3 strip (the C operand of its S chains), 4..11 row fragments
    auto next_read = [&](int idx) {
#ifndef QDE2_ABL_NOROWS
      if (idx < 4) {
        const f32x4 t4 = *(const lds_f4*)(bs + 128 * NIT + 32 * idx);
        sv[4 * idx + 0] = t4[0];
        sv[4 * idx + 1] = t4[1];
        sv[4 * idx + 2] = t4[2];
        sv[4 * idx + 3] = t4[3];
      } else {
        const int sa = idx - 4;
        af[sa] = *(const lds_bf16x8*)(((sa & 1) ? b1 : b0) + noff + 512 * (sa >> 1));
      }
#endif
    };
    // ---- slot 3
    bf16x8 pa0 = {}, pb0 = {};
    {
#pragma unroll
      for (int m = 0; m < 2 * FT; ++m) {
        const int ft = m % FT, s2 = m / FT;
        if (m == 0) pa0 = frag(pw0, 0);
        if (m == FT) pb0 = frag(pw0, 1);
        y[0][ft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[ft][s2], s2 ? pb0 : pa0, y[0][ft], 0, 0, 0);
        QDE2_FENCE();
        if (m < LAG) chunk(acc0, 0, KS - LAG + m, pw0, c0, left);   // chunks 5 .. 7 of P0
        else chunk(acc1, 1, m - LAG, pw1, c1, left);                // chunks 0 .. 4 of P1
        QDE2_FENCE();
        next_read(m);                                               // one read behind each MFMA: strip, rows 0 .. 3
        QDE2_FENCE();
      }
    }
    cs[0] += c0;
    // ---- slot 4
    bf16x8 pa1 = {}, pb1 = {};
    auto de1 = [&](int m) {
      const int ft = m % FT, s2 = m / FT;
      y[1][ft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[ft][s2], s2 ? pb1 : pa1, y[1][ft], 0, 0, 0);
      QDE2_FENCE();
    };
    pa1 = frag(pw1, 0);
#pragma unroll
    for (int m = 0; m < LAG; ++m) {
      de1(m);
      chunk(acc1, 1, KS - LAG + m, pw1, c1, left);              // chunks 5 .. 7 of P1
      QDE2_FENCE();
    }
    cs[1] += c1;
    pb1 = frag(pw1, 1);
#pragma unroll
    for (int m = LAG; m < 2 * FT; ++m) {
      de1(m);
      if (2 * FT + (m - LAG) < KS + 4) {
        next_read(2 * FT + (m - LAG));                             // rows 4 .. 7
        QDE2_FENCE();
      }
    }
  };

  load_owner(g);

  // ---- prologue of the ring: stages 0 and 1 in flight, rows of the first tile in registers ---------------------------
  int issued = 0;
  for (int s0 = 0; s0 < 2 && s0 < nst; ++s0) {
    issue(t_dma, s0);
    ++issued;
    if (++t_dma == a.T) t_dma = 0;
  }
  de_wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  using I0 = std::integral_constant<int, 0>;
  using BT = std::integral_constant<bool, true>;
  using BF = std::integral_constant<bool, false>;
  read_rows(BF{}, I0{});

  // pieces of item groups (outer) x stages of the piece (inner): the owner fragments are invariant in the inner loop
  int j = 0, cur_buf = 0;
  while (j < nst) {
    int seg_end = j + (a.T - t);
    if (seg_end > nst) seg_end = nst;
    for (; j < seg_end; ++j) {
      const bool more = j + 1 < nst, refill = issued < nst;
      if constexpr (C::TILES == 2) tile(I0{}, BF{}, true, false, cur_buf);
      tile(std::integral_constant<int, C::TILES - 1>{}, BT{}, more, refill, cur_buf);
      if (more && refill) ++issued;
      ++t;
      // the ring turned: the other buffer is the current one now
      { const lds_u8* x = pA0; pA0 = nA0; nA0 = x; }
      { const lds_u8* x = pA1; pA1 = nA1; nA1 = x; }
      { const lds_u8* x = pT0; pT0 = nT0; nT0 = x; }
      { const lds_u8* x = pT1; pT1 = nT1; nT1 = x; }
      { const lds_u8* x = pS; pS = nS; nS = x; }
      cur_buf ^= 1;
    }
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
      a.stamps[2 * blockIdx.x + 1] = rt - stamp_rt;
    }
  }
}

// =============================================================================================================
// host side
// =============================================================================================================
template <int D, bool MASK>
static void qde2_launch_n(const QDeArgs& a, int grid, hipStream_t s) {
  constexpr int smem = 2 * DeCfg<D, 4>::BUF_BYTES;
  hipLaunchKernelGGL((qde2_kernel<D, MASK>), dim3(grid), dim3(256), smem, s, a);
}

// rows [0, n_items): `a` prepared by cql_qde_launch (G, T for 256-item groups; nlse2 = -lse in NATURAL units here)
int cql_qde2_run(const QDeArgs& a, int d, int grid, hipStream_t s) {
  const bool mask = (a.n_states % DeCfg<128, 4>::TI) != 0;
  if (d == 128) { if (mask) qde2_launch_n<128, true>(a, grid, s); else qde2_launch_n<128, false>(a, grid, s); }
  else return CQLREC_ERR_INVALID;
  return CQLREC_OK;
}
