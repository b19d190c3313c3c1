// Fused forward of the training step ("flash" form: logsumexp AND the softmax-weighted item sum in one catalogue pass),
// second form (d = 128): ONE wave per SIMD, each wave owns TWO 32-state groups (64 states, 256 per block), software-
// pipelined inside the wave -- qde2_kernel (qhead_de2.hip) with the roles swapped: the owners are the STATES, the item
// table streams through LDS, the strip carries the items' bias and is the C operand of the score chains.
//
// Why: the QM_LSE_DH mode of qstream_kernel (32 states per wave, two waves per SIMD) re-reads every item tile from LDS --
// 8 row reads + 16 transposed reads + the strip = 28 KB -- for only 16 MFMAs; with all waves of a CU reading in step
// that is ~220 B/clk of LDS traffic, the LDS limit (0.207 ms at cfg3, MFMA busy 43 %).  Here the same 28 KB feed 32 MFMAs.
//
// Reference of the exponentials: FIXED per (item slice, state) -- the maximum of the slice's FIRST tile plus
// QF2_REF_MARGIN nats -- instead of the running reference of the first form.  P = exp(S - ref) then exceeds 1 for scores
// above the reference, which costs nothing (bf16 and fp32 keep their relative precision over 2^+-126; the partial sums
// are merged relative to the slices' references by qhead_finalize_lse_kernel / qhead_dh_finish_kernel as before); only a
// score more than ~80 nats above the first tile's maximum would overflow.  That case is detected (non-finite partial
// sum -> flag) and the caller's next launch, the first form guarded by the flag, recomputes the pass exactly.  No
// rescaling of the 128 accumulator registers, no per-tile maximum, no branch in the loop.
//
// Per tile (32 items x this wave's 2 x 32 states) the period of qde2_kernel: chains A = S0, B = S1 (scores of the two state
// groups), C = Y0, D = Y1 (Y += E^T . P), 32 half-chunks of the exponentials, one LDS read per gap (see qhead_de2.hip).
#include <stdlib.h>
#include <type_traits>
#include "qhead_de_common.h"

#define QF2_REF_MARGIN 8.0f

// Y += A . B with the accumulator in AccVGPRs, as inline asm: the builtin form of this translation unit
// (-amdgpu-mfma-vgpr-form, which the score chains need: the VALU reads their results) would put the 128 accumulator
// registers of Y into the VGPR half too, and the rest of the kernel then no longer fits there -- hipcc parks owner
// fragments in AccVGPRs and copies them back in front of every product.  A = transposed item fragment (VGPRs: where
// hipcc lets the ds_read_tr land), B = probability fragment (VGPRs).  The leading s_nop covers "VALU wrote B just before"
// (the hazard recogniser does not look into asm); products on the same accumulator are four products apart.
#ifndef QF2_BUILTIN_Y
__device__ __forceinline__ void qf2_mfma_y(f32x16& y, const bf16x8& a_frag, const bf16x8& b_frag) {
  const u32x4 av = __builtin_bit_cast(u32x4, a_frag), bv = __builtin_bit_cast(u32x4, b_frag);
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(y) : "v"(av), "v"(bv));
}
#else
__device__ __forceinline__ void qf2_mfma_y(f32x16& y, const bf16x8& a_frag, const bf16x8& b_frag) {
  y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_frag, b_frag, y, 0, 0, 0);
}
#endif

#ifdef QF2_PROBE_PSTORE
__device__ uint32_t* qf2_probe_p_dev;      // timing-only probe buffer (cql_qfwd2_run allocates it)
#endif
template <int D>
__global__ __launch_bounds__(256, 1) void qfwd2_kernel(QFwd2Args a) {
  using C = DeCfg<D, 4>;
  constexpr int KS = C::KS, FT = C::FT;
  static_assert(C::TILES == 2, "two tiles per stage");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // the ONLY LDS object of this kernel

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int split = blockIdx.x % a.nsplit;
  const int64_t rblk = blockIdx.x / a.nsplit;
  const int64_t s_begin = (int64_t)split * a.split_rows;
  const int64_t s_end = (s_begin + a.split_rows < a.n_items) ? (s_begin + a.split_rows) : a.n_items;
  const int nst = (s_end > s_begin) ? (int)((s_end - s_begin + C::TI - 1) / C::TI) : 0;
#ifdef QF2_PROBE_PSTORE
  uint32_t* const probe_p = qf2_probe_p_dev;
#endif
  if (nst <= 0) return;
  const uint32_t gst0 = (uint32_t)(s_begin / C::TI);

  // ---- staging (see qde_kernel) ----------------------------------------------------------------------------------
  __amdgpu_buffer_rsrc_t rs_e = __builtin_amdgcn_make_buffer_rsrc((void*)a.E_b, 0, (int)(a.n_items * C::ROWB), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, (int)(a.n_items * 4), 0x00020000);
  uint32_t voff;
  {
    const int sub = lane >> 5, r7 = (lane >> 2) & 7, slot = lane & 3;
    const int rg0 = wave / C::PPG, hc = wave % C::PPG;
    const int q2 = (r7 >> 2) | ((rg0 & 1) << 1);
    voff = (uint32_t)((rg0 * 8 + r7) * C::ROWB + (8 * hc + 4 * sub + (slot ^ q2)) * 16);
  }
  const uint32_t voff_strip = (uint32_t)lane * 4;
  const uint32_t smem_base = lds_addr_of(smem);
  auto issue = [&](int stage, int buf) __attribute__((always_inline)) {
    const uint32_t bufp = __builtin_amdgcn_readfirstlane(smem_base + buf * C::BUF_BYTES);
    const uint32_t gs = gst0 + (uint32_t)stage;
#pragma unroll
    for (int i = 0; i < C::LPS; ++i) bdma16(voff, rs_e, gs * C::STAGE_BYTES + C::PSTEP * i, bufp + (4 * i + wave) * 1024);
    if (wave == (stage & 3)) bdma4(voff_strip, rs_b, gs * (C::TI * 4), bufp + C::STAGE_BYTES);
  };
  // Items past the end of the catalogue (last stage of the last slice): their rows and bias read as 0 (buffer bounds);
  // a bias of -inf makes their scores -inf and their probabilities 0.  Block-uniform; between two barriers.
  auto patch_strip = [&](int stage, int buf) __attribute__((always_inline)) {
    const int64_t valid = a.n_items - (int64_t)(gst0 + (uint32_t)stage) * C::TI;
    if (valid < C::TI) {
      if (wave == 0 && lane >= valid)
        *(__attribute__((address_space(3))) float*)((lds_u8*)smem + buf * C::BUF_BYTES + C::STAGE_BYTES + lane * 4) = NEG_INF_F;
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
    }
  };

  // ---- read geometry (see qde2_kernel): two sets of per-lane bases, swapped at the end of every stage -------------------
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
  auto swap_bufs = [&]() __attribute__((always_inline)) {
    { const lds_u8* x = pA0; pA0 = nA0; nA0 = x; }
    { const lds_u8* x = pA1; pA1 = nA1; nA1 = x; }
    { const lds_u8* x = pT0; pT0 = nT0; nT0 = x; }
    { const lds_u8* x = pT1; pT1 = nT1; nT1 = x; }
    { const lds_u8* x = pS; pS = nS; nS = x; }
  };

  // ---- owner state: two 32-state groups per wave -------------------------------------------------------------------------
  bf16x8 rf[2][KS];
  f32x16 y[2][FT];
  float cs[2] = {0.f, 0.f};     // running sums of P (this lane's 16 rows of every tile)
  float rl2[2];                 // -reference * log2e of this lane's state in each group
  // fragments of state group `grp` into rf[slot]; retired (in hipcc's own bookkeeping too) before the next LDS-DMA
  auto load_owner = [&](int slot, int grp, const uint16_t* base) __attribute__((always_inline)) {
    int64_t row = rblk * 256 + wave * 64 + grp * 32 + r;
    if (row >= a.n_states) row = a.n_states - 1;
#pragma unroll
    for (int s = 0; s < KS; ++s) rf[slot][s] = *reinterpret_cast<const bf16x8*>(base + row * D + 16 * s + 8 * h);
  };
  auto owner_fence = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_s_waitcnt(0x0F70);       // (see qde2_kernel::load_owner)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
#pragma unroll
  for (int gi = 0; gi < 2; ++gi)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int i = 0; i < 16; ++i) y[gi][ft][i] = 0.f;

#define QF2_FENCE() __builtin_amdgcn_sched_barrier(0)
  bf16x8 af[KS];            // row fragments and bias of ONE tile: those of tile t+1 replace tile t's one by one, each right
  f32x16 sv;                // behind its last use (B(t) reads row s in gap 5+s and the bias in gap 5).  hipcc renames them
  f32x16 acc0, acc1;        // into two register sets anyway where it has room; the source asks for one.
  bf16x8 tf[2][FT][2];
  bf16x8 dpa, dpb;

  // half-chunks of the exponentials (qde2_kernel's; volatile asm: hipcc would otherwise regroup them).  -DQF2_PACKED
  // selects packed forms (v_pk_fma_f32 for both exponent arguments, v_pk_add_f32 into a pair of partial sums: 5 instead
  // of 7 instructions per pair) -- correct, but slower beside MFMAs (0.266 vs 0.198 ms): kept for A/B only.
#ifndef QF2_PACKED
  typedef float csum_t;
  float ht0 = 0.f, ht1 = 0.f;
  auto half_a = [&](const f32x16& acc, int k, float b0) __attribute__((always_inline)) {
    asm volatile(
        "v_fmamk_f32 %0, %2, 0x3fb8aa3b, %4\n\t"
        "v_fmamk_f32 %1, %3, 0x3fb8aa3b, %4\n\t"
        "v_exp_f32 %0, %0"
        : "=&v"(ht0), "=&v"(ht1)
        : "v"(acc[2 * k]), "v"(acc[2 * k + 1]), "v"(b0));
  };
  auto half_b = [&](uint32_t& w, float& csum) __attribute__((always_inline)) {
    asm volatile(
        "v_exp_f32 %1, %1\n\t"
        "v_add_f32 %3, %3, %0\n\t"
        "v_add_f32 %3, %3, %1\n\t"
        "v_cvt_pk_bf16_f32 %2, %0, %1"
        : "+v"(ht0), "+v"(ht1), "=&v"(w), "+v"(csum));
  };
  auto csum_total = [](float c) __attribute__((always_inline)) { return c; };
#else
  // packed form: v_pk_fma_f32 forms both exponent arguments (accumulator elements 2k, 2k+1 are a register pair),
  // v_pk_add_f32 adds both values to a pair of partial sums (even / odd elements)
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  typedef f32x2 csum_t;
  f32x2 ht = {0.f, 0.f};
  f32x2 l2e2 = {CQL_LOG2E, CQL_LOG2E};
  asm volatile("" : "+v"(l2e2));
  auto half_a = [&](const f32x16& acc, int k, float b0) __attribute__((always_inline)) {
    const f32x2 a2 = {acc[2 * k], acc[2 * k + 1]};
    const f32x2 b2 = {b0, b0};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,1]" : "=&v"(ht) : "v"(a2), "v"(l2e2), "v"(b2));
    float e0 = ht[0];
    asm volatile("v_exp_f32 %0, %0" : "+v"(e0));
    ht[0] = e0;
  };
  auto half_b = [&](uint32_t& w, f32x2& csum) __attribute__((always_inline)) {
    float e1 = ht[1];
    asm volatile("v_exp_f32 %0, %0" : "+v"(e1));
    ht[1] = e1;
    asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,1]" : "+v"(csum) : "v"(ht));
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(ht[0]), "v"(ht[1]));
  };
  auto csum_total = [](f32x2 c) __attribute__((always_inline)) { return c[0] + c[1]; };
#endif
  auto frag = [](const uint32_t (&pw)[8], int s2) __attribute__((always_inline)) {
    u32x4 v = {pw[4 * s2 + 0], pw[4 * s2 + 1], pw[4 * s2 + 2], pw[4 * s2 + 3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  // LDS reads of the tile FOLLOWING tile (cur buffer, IT): kind 0 = bias quarter idx, 1 = row fragment idx, 2 = transposed
  // read idx (0..15)
  auto next_read = [&](auto IT, int kind, int idx) __attribute__((always_inline)) {
    constexpr bool END = decltype(IT)::value == C::TILES - 1;
    constexpr int NIT = END ? 0 : decltype(IT)::value + 1;
    constexpr int noff = NIT * C::TILE_BYTES;
    if (kind == 0) {
      const f32x4 t4 = *(const lds_f4*)((END ? nS : pS) + 128 * NIT + 32 * idx);
      sv[4 * idx + 0] = t4[0];
      sv[4 * idx + 1] = t4[1];
      sv[4 * idx + 2] = t4[2];
      sv[4 * idx + 3] = t4[3];
    } else if (kind == 1) {
      af[idx] = *(const lds_bf16x8*)(((idx & 1) ? (END ? nA1 : pA1) : (END ? nA0 : pA0)) + noff + 512 * (idx >> 1));
    } else {
      const int ft = idx >> 2, s2 = (idx >> 1) & 1, jj = idx & 1;
      const bf16x4 t4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
          (lds_bf16x4*)((jj ? (END ? nT1 : pT1) : (END ? nT0 : pT0)) + noff + C::RG_BYTES * (2 * s2 + jj) + 512 * ft));
      tf[NIT & 1][ft][s2][4 * jj + 0] = t4[0];
      tf[NIT & 1][ft][s2][4 * jj + 1] = t4[1];
      tf[NIT & 1][ft][s2][4 * jj + 2] = t4[2];
      tf[NIT & 1][ft][s2][4 * jj + 3] = t4[3];
    }
  };
  // one read per gap from gap 4 on: transposed 0, 1 | rows 0..7 (gap 6 + s: right behind B's product s) | bias 0..3 |
  // transposed 2..15
  auto gap_read = [&](auto IT, int gp) __attribute__((always_inline)) {
    if (gp < 4) return;
    if (gp < 6) next_read(IT, 2, gp - 4);
    else if (gp < 14) next_read(IT, 1, gp - 6);
    else if (gp < 18) next_read(IT, 0, gp - 14);
    else next_read(IT, 2, gp - 16);
  };
  int st = 0, issued = 0, cur_buf = 0;
  // the ring turns (tile = last of its stage): the next stage's pieces have landed for everyone, everyone has left this
  // stage's buffer (its last reads were issued in the previous period), which is refilled with stage + 2
  auto ring_turn = [&]() __attribute__((always_inline)) {
    if (st + 1 < nst) {
      __builtin_amdgcn_s_waitcnt(0xC07F);
      de_wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      patch_strip(st + 1, cur_buf ^ 1);
      if (issued < nst) {
        issue(issued, cur_buf);
        ++issued;
      }
    }
  };
  // one period (qde2_kernel::period): D(t-1) last 5 | B(t) | C(t) | A(t+1) | D(t) first 3
  auto period = [&](auto IT) __attribute__((always_inline)) {
    constexpr int P = decltype(IT)::value & 1;
    constexpr bool END = decltype(IT)::value == C::TILES - 1;
    uint32_t pw0[8], pw1[8];
    csum_t c0 = {}, c1 = {};
    bf16x8 pa0 = {}, pb0 = {}, pa1 = {};
#pragma unroll
    for (int gp = 0; gp < 32; ++gp) {
      if (gp < 5) {
        const int m = 3 + gp, ft = m % FT, s2 = m / FT;
        qf2_mfma_y(y[1][ft], tf[P ^ 1][ft][s2], s2 ? dpb : dpa);
      } else if (gp < 13) {
        const int s = gp - 5;
        if (s == 0) acc1 = sv;
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], rf[1][s], acc1, 0, 0, 0);
      } else if (gp < 21) {
        const int m = gp - 13, ft = m % FT, s2 = m / FT;
        if (m == 0) pa0 = frag(pw0, 0);
        if (m == FT) pb0 = frag(pw0, 1);
        qf2_mfma_y(y[0][ft], tf[P][ft][s2], s2 ? pb0 : pa0);
      } else if (gp < 29) {
        const int s = gp - 21;
        if (s == 0) acc0 = sv;
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], rf[0][s], acc0, 0, 0, 0);
      } else {
        const int m = gp - 29, ft = m % FT;
        if (m == 0) pa1 = frag(pw1, 0);
        qf2_mfma_y(y[1][ft], tf[P][ft][0], pa1);
      }
      QF2_FENCE();
      {
        const int hc = gp & 15, k = hc >> 1;
        if (gp < 16) {
          if ((hc & 1) == 0) half_a(acc0, k, rl2[0]);
          else half_b(pw0[k], c0);
        } else {
          if ((hc & 1) == 0) half_a(acc1, k, rl2[1]);
          else half_b(pw1[k], c1);
        }
      }
      QF2_FENCE();
      if (gp == 4) {
        if constexpr (END) ring_turn();
      }
      gap_read(IT, gp);
      QF2_FENCE();
    }
    cs[0] += csum_total(c0);
    cs[1] += csum_total(c1);
    dpa = pa1;
    dpb = frag(pw1, 1);
#ifdef QF2_PROBE_PSTORE      // timing-only probe: what does it cost this kernel to write its P tiles out?
    {                        // 2 groups x 32 B per lane and tile, as four 1 KiB wave stores into a per-wave stream
      const int64_t tile = ((int64_t)blockIdx.x * 4 + wave) * (2 * nst) + 2 * st + decltype(IT)::value;
      uint32_t* dst = probe_p + tile * 1024 + lane * 4;
      *reinterpret_cast<u32x4*>(dst) = u32x4{pw0[0], pw0[1], pw0[2], pw0[3]};
      *reinterpret_cast<u32x4*>(dst + 256) = u32x4{pw0[4], pw0[5], pw0[6], pw0[7]};
      *reinterpret_cast<u32x4*>(dst + 512) = u32x4{pw1[0], pw1[1], pw1[2], pw1[3]};
      *reinterpret_cast<u32x4*>(dst + 768) = u32x4{pw1[4], pw1[5], pw1[6], pw1[7]};
    }
#endif
  };

#pragma unroll
  for (int pp = 0; pp < 2; ++pp)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) tf[pp][ft][s2] = bf16x8{};     // the first tile's pending-D products add 0 * this
  dpa = bf16x8{};
  dpb = bf16x8{};

  // ---- prologue: stages 0 and 1 in flight; rows, strip and transposed fragments of the first tile in registers ----
  for (int s0 = 0; s0 < 2 && s0 < nst; ++s0) {
    issue(s0, s0);
    ++issued;
  }
  de_wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  patch_strip(0, 0);
  swap_bufs();      // "the tile following tile -1": next_read with the roles of the buffers swapped
#pragma unroll
  for (int gp = 4; gp < 32; ++gp) gap_read(std::integral_constant<int, C::TILES - 1>{}, gp);
  swap_bufs();
  // Scores of the first tile for both groups, through TEMPORARY fragments: their maxima fix the references.  (The
  // fragments the loop keeps, rf, are loaded afterwards and used by the loop only -- plus A of tile 0, as in qde2_kernel:
  // with one more use in front of the loop hipcc rotates them through AccVGPR tuples, four copies per product.)
  float ref_a[2];       // the references, parked in AccVGPRs until the end (the loop needs only rl2)
#pragma unroll
  for (int gi = 0; gi < 2; ++gi) {
    bf16x8 tmpf[KS];
    {
      int64_t row = rblk * 256 + wave * 64 + gi * 32 + r;
      if (row >= a.n_states) row = a.n_states - 1;
#pragma unroll
      for (int s = 0; s < KS; ++s) tmpf[s] = *reinterpret_cast<const bf16x8*>(a.H_b + row * D + 16 * s + 8 * h);
    }
    f32x16 t = sv;
#pragma unroll
    for (int s = 0; s < KS; ++s) t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], tmpf[s], t, 0, 0, 0);
    float m = t[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) m = fmaxf(m, t[i]);
    m = fmaxf(m, __shfl_xor(m, 32));
    const float rv = (m == NEG_INF_F) ? 0.f : m + QF2_REF_MARGIN;
    rl2[gi] = -rv * CQL_LOG2E;
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ref_a[gi]) : "v"(rv));
    asm volatile("" : "+v"(rl2[gi]));      // (keeps the temporaries' uses in front of the loads below)
  }
  load_owner(0, 0, a.H_b);
  load_owner(1, 1, a.H_b);
  owner_fence();
  acc0 = sv;        // A of the first tile (its rows are in registers)
#pragma unroll
  for (int s = 0; s < KS; ++s) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], rf[0][s], acc0, 0, 0, 0);

  for (st = 0; st < nst; ++st) {
    period(std::integral_constant<int, 0>{});
    period(std::integral_constant<int, 1>{});
    swap_bufs();
    cur_buf ^= 1;
  }
  // the last 5 products of D of the last tile
#pragma unroll
  for (int m = 3; m < 2 * FT; ++m) {
    const int ft = m % FT, s2 = m / FT;
    qf2_mfma_y(y[1][ft], tf[1][ft][s2], s2 ? dpb : dpa);
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last products have left the pipe before Y is read

  // ---- partials: (reference, sum relative to it) and the un-normalised slab -----------------------------------------
#pragma unroll
  for (int gi = 0; gi < 2; ++gi) {
    const int64_t row = rblk * 256 + wave * 64 + gi * 32 + r;
    const float ls = cs[gi] + __shfl_xor(cs[gi], 32);
    if (row < a.n_states) {
      const int64_t pidx = (int64_t)split * a.n_states + row;
      float* dst = a.slab + pidx * D;
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(dst + ft * 32 + 8 * q + 4 * h) =
              make_float4(y[gi][ft][4 * q + 0], y[gi][ft][4 * q + 1], y[gi][ft][4 * q + 2], y[gi][ft][4 * q + 3]);
      if (h == 0) {
        float rv;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(rv) : "a"(ref_a[gi]));
        a.part_a[pidx] = rv;
        a.part_b[pidx] = ls;
        if (!(ls < 3.0e38f) && a.flag) atomicOr(a.flag, 1);      // inf or NaN: the guarded first form redoes the pass
      }
    }
  }
}

// =============================================================================================================
// host side
// =============================================================================================================
bool cql_qfwd2_supported(int d, int64_t n_items) {
  static const int off = getenv("CQL_QFWD2") && getenv("CQL_QFWD2")[0] == '0';
  return !off && d == 128 && n_items * 256 < (1ll << 31);
}

int cql_qfwd2_run(const QFwd2Args& a, int d, hipStream_t s) {
  if (!cql_qfwd2_supported(d, a.n_items)) return CQLREC_ERR_INVALID;
#ifdef QF2_PROBE_PSTORE
  {
    static void* buf = nullptr;
    static int64_t cap = 0;
    const int64_t need = ((a.n_states + 255) / 256 * 256) * ((a.n_items + 63) / 64 * 64 + 64 * a.nsplit) * 2 + (1 << 20);
    if (need > cap) {
      if (buf) (void)hipFree(buf);
      if (hipMalloc(&buf, (size_t)need) != hipSuccess) return CQLREC_ERR_HIP;
      cap = need;
      (void)hipMemcpyToSymbol(HIP_SYMBOL(qf2_probe_p_dev), &buf, sizeof(buf));
    }
  }
#endif
  constexpr int smem = 2 * DeCfg<128, 4>::BUF_BYTES;
  const int64_t rblks = (a.n_states + 255) / 256;
  hipLaunchKernelGGL((qfwd2_kernel<128>), dim3((unsigned)(rblks * a.nsplit)), dim3(256), smem, s, a);
  CQL_LAUNCH_CHECK("qfwd2");
  return CQLREC_OK;
}
