// Item-side backward of the Q-head for d = 256: the one-wave-per-SIMD, software-pipelined sibling of qde2_kernel
// (qhead_de2.hip, d = 128), with the period of qfwd3_kernel (qhead_fwd3.hip) and the roles swapped: the owners are the ITEMS
// (32 per wave, 128 per group -- the gradient rows of one group already fill 128 accumulator registers), the states stream
// through LDS (one 32-state tile = one stage), the strip carries -lse in natural units and is the C operand of the score
// chain, the item bias enters the exponent as a per-lane constant:
//
//   dE^T[f][item] = sum_state H_b^T[f][state] * bf16(P[state][item]),   P = exp2((S - lse_state) log2e + b_item log2e)
//
//   gap      MFMA                                    VALU                      LDS
//   0-15     S(t+1) = -lse + H(t+1) . E^T  (16)       P(t): half-chunk per gap   32 transposed reads of tile t (2 per gap)
//   16       -- the ring turns: stage t+2 has landed for everyone, everyone has read tile t; its buffer takes stage t+3 --
//   16-31    dE += H(t)^T . P(t)           (16)       --                         rows + strip of tile t+2; gaps 16-20: LDS-DMA
//
// Work decomposition (persistent blocks over contiguous, equal ranges of the group x stage grid), the two-piece sum of a
// cut group, its slab and the fix-up: exactly those of qde_kernel / qde2_kernel (qhead_de.hip), with which this kernel
// shares group size (128 items), stage size (32 states) and grid -- it replaces qde_kernel<256> launch for launch.
// A piece starts by re-reading its first tile's rows (the pipeline has already replaced them by the rows two tiles ahead)
// and forming that tile's scores on its own; the scores the previous piece formed for it with ITS items are dropped.
#include <stdlib.h>
#include <type_traits>
#include "qhead_de_common.h"

#define QDE3_ITEMS 128
#define QDE3_NBUF 3

__device__ __forceinline__ float qde3_axpy(float w, float scale, float u) {
#pragma clang fp contract(off)
  const float su = scale * u;        // product, then sum: the terms and order of the fix-up paths (see qde2_axpy)
  return w + su;
}

template <int D>
__global__ __launch_bounds__(256, 1) void qde3_kernel(QDeArgs a) {
  using C = DeCfg<D, 4>;
  constexpr int KS = C::KS, FT = C::FT;
  static_assert(D == 256 && C::TILES == 1 && C::LPS == 4 && C::PPG == 4 && C::ITEMS == QDE3_ITEMS, "d = 256 geometry");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // the ONLY LDS object of this kernel

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  const int64_t W = (int64_t)a.G * a.T;
  const int64_t u0 = (int64_t)blockIdx.x * W / gridDim.x, u1 = ((int64_t)blockIdx.x + 1) * W / gridDim.x;
  const int nst = (int)(u1 - u0);
  if (nst <= 0) return;
  int g = (int)(u0 / a.T);
  int t = (int)(u0 - (int64_t)g * a.T);
  int t_seg = t;
  int t_dma = t;

  // ---- staging (qfwd3_kernel's: piece 4 i + wave = 8-row group i, column octet `wave`; parity of the row group = i & 1) ----
  __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)a.H_b, 0, (int)(a.n_states * C::ROWB), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)a.nlse2, 0, (int)(a.n_states * 4), 0x00020000);
  uint32_t voff[2];
  {
    const int sub = lane >> 5, r7 = (lane >> 2) & 7, slot = lane & 3;
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int q2 = (r7 >> 2) | (par << 1);
      voff[par] = (uint32_t)(r7 * C::ROWB + (8 * wave + 4 * sub + (slot ^ q2)) * 16);
    }
  }
  const uint32_t voff_strip = (uint32_t)lane * 4;
  const uint32_t smem_base = lds_addr_of(smem);
  auto issue_piece = [&](int stage_t, int buf, int i) __attribute__((always_inline)) {
    const uint32_t bufp = __builtin_amdgcn_readfirstlane(smem_base + buf * C::BUF_BYTES);
    bdma16(voff[i & 1], rs_h, (uint32_t)stage_t * C::STAGE_BYTES + C::PSTEP * i, bufp + (4 * i + wave) * 1024);
  };
  auto issue_strip = [&](int stage_t, int buf) __attribute__((always_inline)) {
    const uint32_t bufp = __builtin_amdgcn_readfirstlane(smem_base + buf * C::BUF_BYTES);
    if (wave == (stage_t & 3)) bdma4(voff_strip, rs_s, (uint32_t)stage_t * (C::TI * 4), bufp + C::STAGE_BYTES);
  };

  // ---- read geometry ------------------------------------------------------------------------------------------------
  const lds_u8* lbase = (const lds_u8*)smem;
  int oa0, oa1, ot0, ot1, os;
  {
    const int g1 = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
    oa0 = C::RG_BYTES * (r >> 3) + 64 * (r & 7) + 16 * ((0 + h) ^ ((r >> 2) & 3));
    oa1 = C::RG_BYTES * (r >> 3) + 64 * (r & 7) + 16 * ((2 + h) ^ ((r >> 2) & 3));
    ot0 = 64 * (4 * h + q) + 16 * ((2 * g1 + (p >> 1)) ^ ((0 + h) & 3)) + 8 * (p & 1);
    ot1 = 64 * (4 * h + q) + 16 * ((2 * g1 + (p >> 1)) ^ ((2 + h) & 3)) + 8 * (p & 1);
    os = C::STAGE_BYTES + 16 * h;
  }
  const lds_u8 *pT0, *pT1;              // transposed reads: the CURRENT tile's buffer
  const lds_u8 *fA0, *fA1, *fS;         // rows + strip: the buffer of the tile after next (or whichever set_ptrs names)
  auto set_ptrs = [&](int b_cur, int b_far) __attribute__((always_inline)) {
    pT0 = lbase + b_cur * C::BUF_BYTES + ot0;
    pT1 = lbase + b_cur * C::BUF_BYTES + ot1;
    fA0 = lbase + b_far * C::BUF_BYTES + oa0;
    fA1 = lbase + b_far * C::BUF_BYTES + oa1;
    fS = lbase + b_far * C::BUF_BYTES + os;
  };

  // ---- owner state: one 32-item group per wave ---------------------------------------------------------------------------
  bf16x8 rf[KS];
  float bl2;                // bias * log2e of this lane's item
  f32x16 y[FT];
  float cs;
  auto load_owner = [&](int grp) __attribute__((always_inline)) {
    int64_t row = (int64_t)grp * QDE3_ITEMS + wave * 32 + r;
    if (row >= a.n_items) row = a.n_items - 1;
#pragma unroll
    for (int s = 0; s < KS; ++s) rf[s] = *reinterpret_cast<const bf16x8*>(a.E_b + row * D + 16 * s + 8 * h);
    const float bv = a.bias[row];
    __builtin_amdgcn_s_waitcnt(0x0F70);       // (see qde2_kernel::load_owner)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bl2 = bv * CQL_LOG2E;
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int i = 0; i < 16; ++i) y[ft][i] = 0.f;
    cs = 0.f;
  };
  auto store_piece = [&](int grp, bool first) __attribute__((always_inline)) {
    const int64_t row = (int64_t)grp * QDE3_ITEMS + wave * 32 + r;
    const bool ok = row < a.n_items;
    const float csum = cs + __shfl_xor(cs, 32);
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
              o = make_float4(qde3_axpy(old.x, a.scale, y[ft][4 * q + 0]), qde3_axpy(old.y, a.scale, y[ft][4 * q + 1]),
                              qde3_axpy(old.z, a.scale, y[ft][4 * q + 2]), qde3_axpy(old.w, a.scale, y[ft][4 * q + 3]));
            } else {
              o = make_float4(a.scale * y[ft][4 * q + 0], a.scale * y[ft][4 * q + 1], a.scale * y[ft][4 * q + 2],
                              a.scale * y[ft][4 * q + 3]);
            }
            *pd = o;
          }
        if (h == 0) a.out_cs[row] = a.accumulate ? qde3_axpy(a.out_cs[row], a.scale, csum) : a.scale * csum;
      }
    } else {
      const int64_t srow = (int64_t)blockIdx.x * QDE3_ITEMS + wave * 32 + r;
      float* dst = a.slab + srow * D;
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(dst + ft * 32 + 8 * q + 4 * h) =
              make_float4(y[ft][4 * q + 0], y[ft][4 * q + 1], y[ft][4 * q + 2], y[ft][4 * q + 3]);
      if (h == 0) a.slab_cs[srow] = csum;
    }
  };

#define QDE3_FENCE() __builtin_amdgcn_sched_barrier(0)
  bf16x8 af[KS];            // row fragments of the tile whose scores are computed next
  f32x16 sv;                // -lse of its states (C operand of the score chain)
  f32x16 acc0, acc1;        // score accumulators by tile parity inside the piece
  bf16x8 tf[FT][2];         // transposed fragments of the current tile

  float ht0 = 0.f, ht1 = 0.f;
  auto half_a = [&](const f32x16& acc, int k) __attribute__((always_inline)) {
    asm volatile(
        "v_fmamk_f32 %0, %2, 0x3fb8aa3b, %4\n\t"
        "v_fmamk_f32 %1, %3, 0x3fb8aa3b, %4\n\t"
        "v_exp_f32 %0, %0"
        : "=&v"(ht0), "=&v"(ht1)
        : "v"(acc[2 * k]), "v"(acc[2 * k + 1]), "v"(bl2));
  };
  auto half_b = [&](uint32_t& w, float& csum) __attribute__((always_inline)) {
    asm volatile(
        "v_exp_f32 %1, %1\n\t"
        "v_add_f32 %3, %3, %0\n\t"
        "v_add_f32 %3, %3, %1\n\t"
        "v_cvt_pk_bf16_f32 %2, %0, %1"
        : "+v"(ht0), "+v"(ht1), "=&v"(w), "+v"(csum));
  };
  auto frag = [](const uint32_t (&pw)[8], int s2) __attribute__((always_inline)) {
    u32x4 v = {pw[4 * s2 + 0], pw[4 * s2 + 1], pw[4 * s2 + 2], pw[4 * s2 + 3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto far_read = [&](int idx) __attribute__((always_inline)) {     // rows (0..15) and strip quarters (16..19) at fA / fS
    if (idx < KS) {
      af[idx] = *(const lds_bf16x8*)(((idx & 1) ? fA1 : fA0) + 512 * (idx >> 1));
    } else {
      const int q = idx - KS;
      const f32x4 t4 = *(const lds_f4*)(fS + 32 * q);
      sv[4 * q + 0] = t4[0];
      sv[4 * q + 1] = t4[1];
      sv[4 * q + 2] = t4[2];
      sv[4 * q + 3] = t4[3];
    }
  };
  auto tr_read = [&](int q) __attribute__((always_inline)) {
    const int s2 = q >> 4, ft = (q >> 1) & 7, jj = q & 1;
    const bf16x4 t4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        (lds_bf16x4*)((jj ? pT1 : pT0) + C::RG_BYTES * (2 * s2 + jj) + 512 * ft));
    tf[ft][s2][4 * jj + 0] = t4[0];
    tf[ft][s2][4 * jj + 1] = t4[1];
    tf[ft][s2][4 * jj + 2] = t4[2];
    tf[ft][s2][4 * jj + 3] = t4[3];
  };

  int b_cur = 0, b_mid = 1, b_far = 2;
  auto rotate = [&]() __attribute__((always_inline)) {
    const int t_ = b_cur;
    b_cur = b_mid;
    b_mid = b_far;
    b_far = t_;
  };
  // the ring turns at gap 16 of a tile: the stage two tiles ahead has landed for everyone, everyone has completed its
  // reads of this tile, whose buffer takes the stage three tiles ahead (the stream of stages wraps at the group's end and
  // simply runs on past the block's range: what it loads there is never used)
  auto ring_turn = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    de_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
  };

  // one period; PAR = parity of the tile inside its piece: its scores are in acc<PAR>, those of the next tile go to the other
  auto period = [&](auto PAR_) __attribute__((always_inline)) {
    constexpr int PAR = decltype(PAR_)::value;
    uint32_t pw[8];
    float c0 = 0.f;
    bf16x8 pa = {}, pb = {};
#pragma unroll
    for (int gp = 0; gp < 32; ++gp) {
      if (gp < 16) {
        if (PAR == 0) {
          if (gp == 0) acc1 = sv;
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[gp], rf[gp], acc1, 0, 0, 0);
        } else {
          if (gp == 0) acc0 = sv;
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[gp], rf[gp], acc0, 0, 0, 0);
        }
        QDE3_FENCE();
        {
          const int k = gp >> 1;
          if ((gp & 1) == 0) half_a(PAR == 0 ? acc0 : acc1, k);
          else half_b(pw[k], c0);
        }
        QDE3_FENCE();
        tr_read(2 * gp);
        tr_read(2 * gp + 1);
        QDE3_FENCE();
      } else {
        const int m = gp - 16, ft = m % FT, s2 = m / FT;
        if (m == 0) {
          ring_turn();
          pa = frag(pw, 0);
          pb = frag(pw, 1);
        }
        y[ft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[ft][s2], s2 ? pb : pa, y[ft], 0, 0, 0);
        QDE3_FENCE();
        if (m < C::LPS) issue_piece(t_dma, b_cur, m);
        else if (m == C::LPS) issue_strip(t_dma, b_cur);
        QDE3_FENCE();
        far_read(m);
        if (m >= 12) far_read(KS + (m - 12));
        QDE3_FENCE();
      }
    }
    cs += c0;
    if (++t_dma == a.T) t_dma = 0;
  };

  load_owner(g);

  // ---- prologue: the whole ring in flight --------------------------------------------------------------------------------
  for (int s0 = 0; s0 < QDE3_NBUF; ++s0) {
#pragma unroll
    for (int i = 0; i < C::LPS; ++i) issue_piece(t_dma, s0, i);
    issue_strip(t_dma, s0);
    if (++t_dma == a.T) t_dma = 0;
  }
  de_wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  // pieces of item groups (outer) x stages of the piece (inner): the owner fragments are invariant in the inner loop
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  int j = 0;
  while (j < nst) {
    int seg_end = j + (a.T - t);
    if (seg_end > nst) seg_end = nst;
    // the piece's first tile: its rows are in the current buffer (the pipeline's registers hold the rows two tiles on);
    // scores on their own, then the rows of the tile behind it, where the first period expects them
    set_ptrs(b_cur, b_cur);
#pragma unroll
    for (int idx = 0; idx < KS + 4; ++idx) far_read(idx);
    acc0 = sv;
#pragma unroll
    for (int s = 0; s < KS; ++s) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], rf[s], acc0, 0, 0, 0);
    set_ptrs(b_cur, b_mid);
#pragma unroll
    for (int idx = 0; idx < KS + 4; ++idx) far_read(idx);
    for (; j + 1 < seg_end; j += 2) {       // two tiles per trip: the accumulator parity is a compile-time constant
      set_ptrs(b_cur, b_far);
      period(I0{});
      rotate();
      set_ptrs(b_cur, b_far);
      period(I1{});
      rotate();
    }
    if (j < seg_end) {
      set_ptrs(b_cur, b_far);
      period(I0{});
      rotate();
      ++j;
    }
    store_piece(g, t_seg == 0);
    if (j < nst) {
      ++g;
      t = 0;
      t_seg = 0;
      load_owner(g);
    }
  }
  de_wait_vmcnt<0>();      // the stages issued past the range have landed before this wave gives its LDS back
}

// =============================================================================================================
// host side
// =============================================================================================================
bool cql_qde3_supported(int d, int64_t batch) {
  static const int off = getenv("CQL_QDE3") && getenv("CQL_QDE3")[0] == '0';
  return !off && d == 256 && batch % 32 == 0;
}

// rows [0, n_items): `a` prepared by cql_qde_launch (G, T for 128-item groups and 32-state stages; nlse2 = -lse in
// NATURAL units here)
int cql_qde3_run(const QDeArgs& a, int d, int grid, hipStream_t s) {
  if (!cql_qde3_supported(d, a.n_states)) return CQLREC_ERR_INVALID;
  constexpr int smem = QDE3_NBUF * DeCfg<256, 4>::BUF_BYTES;
  hipLaunchKernelGGL((qde3_kernel<256>), dim3(grid), dim3(256), smem, s, a);
  return CQLREC_OK;
}
