// Fused forward of the training step (logsumexp AND the softmax-weighted item sum in one catalogue pass) for d = 256:
// the one-wave-per-SIMD, software-pipelined sibling of qfwd2_kernel (qhead_fwd2.hip, d = 128).  Role in the reference's
// structure: the forward of the catalogue-wide output layer and its log-softmax (replay/models/mult_vae.py:101, :275-276 are
// the closest in-tree analogue); arithmetic: SURVEY 8.0 S4/S5 as restated in DESIGN 2.
//
// What is different from d = 128.  The softmax-weighted sum Y of ONE 32-state group already fills 128 accumulator
// registers (8 feature tiles), so a wave owns one group, not two (128 states per block), and a 32-item tile -- 16 KiB of
// LDS -- feeds 32 MFMAs of one wave: twice the LDS fill and read traffic per MFMA of qfwd2, half its exponentials.  The
// period of tile t therefore has two halves instead of four chains:
//
//   gap      MFMA                                   VALU                      LDS
//   0-15     S(t+1) = bias + E(t+1) . H^T  (16)      P(t): half-chunk per gap   32 transposed reads of tile t (2 per gap)
//   16       -- the ring turns: stage t+2 has landed for everyone, everyone has read tile t; its buffer takes stage t+3 --
//   16-31    Y += E(t)^T . P(t)            (16)      --                         rows + bias of tile t+2 (20 reads); gaps 16-20:
//                                                                                the LDS-DMA pieces of stage t+3
//
// One tile = one stage (32 items x 512 B); three ring buffers.  The scores of tile t+1 are computed one period ahead into
// the OTHER accumulator set, so the exponentials of tile t never wait for a matrix product, and Y's products find their
// probability fragments finished 1 (first half) and 9 gaps (second half) earlier.
// Fixed per-(slice, state) reference, overflow flag + guarded fall-back, partial results: exactly as qfwd2_kernel.
#include <stdlib.h>
#include <type_traits>
#include "qhead_de_common.h"

#define QF3_REF_MARGIN 8.0f
#define QF3_NBUF 3

__device__ __forceinline__ void qf3_mfma_y(f32x16& y, const bf16x8& a_frag, const bf16x8& b_frag) {
  const u32x4 av = __builtin_bit_cast(u32x4, a_frag), bv = __builtin_bit_cast(u32x4, b_frag);
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(y) : "v"(av), "v"(bv));
}

template <int D>
__global__ __launch_bounds__(256, 1) void qfwd3_kernel(QFwd2Args a) {
  using C = DeCfg<D, 4>;
  constexpr int KS = C::KS, FT = C::FT;
  static_assert(D == 256 && C::TILES == 1 && C::LPS == 4 && C::PPG == 4, "one 32-item tile per stage, four pieces per wave");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // the ONLY LDS object of this kernel

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int split = blockIdx.x % a.nsplit;
  const int64_t rblk = blockIdx.x / a.nsplit;
  const int64_t s_begin = (int64_t)split * a.split_rows;
  const int64_t s_end = (s_begin + a.split_rows < a.n_items) ? (s_begin + a.split_rows) : a.n_items;
  const int nst = (s_end > s_begin) ? (int)((s_end - s_begin + C::TI - 1) / C::TI) : 0;
  if (nst <= 0) return;
  const uint32_t gst0 = (uint32_t)(s_begin / C::TI);

  // ---- staging: piece 4 i + wave of a stage = 8-row group i, column octet `wave` (see qde_kernel; the row group's parity,
  // which enters the image's swizzle, alternates with i here) ---------------------------------------------------------
  __amdgpu_buffer_rsrc_t rs_e = __builtin_amdgcn_make_buffer_rsrc((void*)a.E_b, 0, (int)(a.n_items * C::ROWB), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, (int)(a.n_items * 4), 0x00020000);
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
  auto issue_piece = [&](int stage, int buf, int i) __attribute__((always_inline)) {
    const uint32_t bufp = __builtin_amdgcn_readfirstlane(smem_base + buf * C::BUF_BYTES);
    const uint32_t gs = gst0 + (uint32_t)stage;
    bdma16(voff[i & 1], rs_e, gs * C::STAGE_BYTES + C::PSTEP * i, bufp + (4 * i + wave) * 1024);
  };
  auto issue_strip = [&](int stage, int buf) __attribute__((always_inline)) {
    const uint32_t bufp = __builtin_amdgcn_readfirstlane(smem_base + buf * C::BUF_BYTES);
    const uint32_t gs = gst0 + (uint32_t)stage;
    if (wave == (stage & 3)) bdma4(voff_strip, rs_b, gs * (C::TI * 4), bufp + C::STAGE_BYTES);
  };
  // Items past the end of the catalogue (last stage of the last slice): rows and bias read as 0 (buffer bounds); a bias of
  // -inf makes their probabilities exactly 0.  Block-uniform; called between the turn's barrier and the first read.
  auto patch_strip = [&](int stage, int buf) __attribute__((always_inline)) {
    const int64_t valid = a.n_items - (int64_t)(gst0 + (uint32_t)stage) * C::TI;
    if (valid < C::TI) {
      if (wave == 0 && lane < C::TI && lane >= valid)
        *(__attribute__((address_space(3))) float*)((lds_u8*)smem + buf * C::BUF_BYTES + C::STAGE_BYTES + lane * 4) = NEG_INF_F;
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
    }
  };

  // ---- read geometry (qde_kernel's image): per-lane offsets inside a buffer; bases recomputed when the ring turns ----------
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
  const lds_u8 *fA0, *fA1, *fS;         // rows + bias: the buffer of the tile after next
  auto set_ptrs = [&](int b_cur, int b_far) __attribute__((always_inline)) {
    pT0 = lbase + b_cur * C::BUF_BYTES + ot0;
    pT1 = lbase + b_cur * C::BUF_BYTES + ot1;
    fA0 = lbase + b_far * C::BUF_BYTES + oa0;
    fA1 = lbase + b_far * C::BUF_BYTES + oa1;
    fS = lbase + b_far * C::BUF_BYTES + os;
  };

  // ---- owner state: one 32-state group per wave ------------------------------------------------------------------------
  bf16x8 rf[KS];
  f32x16 y[FT];
  float cs = 0.f;           // running sum of P (this lane's 16 rows of every tile)
  float rl2;                // -reference * log2e of this lane's state
#pragma unroll
  for (int ft = 0; ft < FT; ++ft)
#pragma unroll
    for (int i = 0; i < 16; ++i) y[ft][i] = 0.f;

#define QF3_FENCE() __builtin_amdgcn_sched_barrier(0)
  bf16x8 af[KS];            // row fragments of the tile whose scores are computed next
  f32x16 sv;                // its bias (C operand of the score chain)
  f32x16 acc0, acc1;        // score accumulators by tile parity
  bf16x8 tf[FT][2];         // transposed fragments of the current tile

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
  auto frag = [](const uint32_t (&pw)[8], int s2) __attribute__((always_inline)) {
    u32x4 v = {pw[4 * s2 + 0], pw[4 * s2 + 1], pw[4 * s2 + 2], pw[4 * s2 + 3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  // rows (idx 0..15) and bias quarters (idx 16..19) of the far tile; transposed read q (0..31) of the current tile
  auto far_read = [&](int idx) __attribute__((always_inline)) {
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
    const int s2 = q >> 4, ft = (q >> 1) & 7, jj = q & 1;      // s2-major: Y's first eight products need s2 = 0
    const bf16x4 t4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        (lds_bf16x4*)((jj ? pT1 : pT0) + C::RG_BYTES * (2 * s2 + jj) + 512 * ft));
    tf[ft][s2][4 * jj + 0] = t4[0];
    tf[ft][s2][4 * jj + 1] = t4[1];
    tf[ft][s2][4 * jj + 2] = t4[2];
    tf[ft][s2][4 * jj + 3] = t4[3];
  };

  int st = 0, issued = 0;
  int b_cur = 0, b_mid = 1 % QF3_NBUF, b_far = 2 % QF3_NBUF;
  // the ring turns at gap 16 of tile st: stage st+2 has landed for everyone and everyone has issued (and, after the
  // lgkmcnt wait, completed) its reads of tile st, whose buffer takes stage st+3
  auto ring_turn = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    de_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    patch_strip(st + 2, b_far);
  };

  // one period; PAR = parity of tile st: its scores are in acc<PAR>, those of tile st+1 go to the other set.  There is ONE
  // form and no branch in it: behind the slice's last tiles the ring keeps turning -- the stages it issues lie past the
  // slice (the buffer descriptor bounds them, nobody reads them: 3 stages of wasted LDS-DMA per slice of ~390), the rows
  // it reads and the scores it forms from them are never used.  (Run-time variants of the period would put the 128
  // accumulator registers of Y under several branches; hipcc then keeps a second set and spills the state fragments.)
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
        QF3_FENCE();
        {
          const int k = gp >> 1;
          if ((gp & 1) == 0) half_a(PAR == 0 ? acc0 : acc1, k, rl2);
          else half_b(pw[k], c0);
        }
        QF3_FENCE();
        tr_read(2 * gp);
        tr_read(2 * gp + 1);
        QF3_FENCE();
      } else {
        const int m = gp - 16, ft = m % FT, s2 = m / FT;
        if (m == 0) {
          ring_turn();
          pa = frag(pw, 0);
          pb = frag(pw, 1);
        }
        qf3_mfma_y(y[ft], tf[ft][s2], s2 ? pb : pa);
        QF3_FENCE();
        // the buffer just left is refilled three stages ahead, one piece per gap behind the turn
        if (m < C::LPS) issue_piece(issued, b_cur, m);
        else if (m == C::LPS) issue_strip(issued, b_cur);
        QF3_FENCE();
        far_read(m);          // rows of the tile after next, one per gap; its bias in the last four gaps
        if (m >= 12) far_read(KS + (m - 12));
        QF3_FENCE();
      }
    }
    cs += c0;
    ++issued;
  };

  // ---- prologue: the whole ring in flight ------------------------------------------------------------------------------
  for (int s0 = 0; s0 < QF3_NBUF && s0 < nst; ++s0) {
#pragma unroll
    for (int i = 0; i < C::LPS; ++i) issue_piece(s0, s0, i);
    issue_strip(s0, s0);
    ++issued;
  }
  de_wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  for (int s0 = 0; s0 < QF3_NBUF && s0 < nst; ++s0) patch_strip(s0, s0);
  // rows + bias of tile 0 (read as "the far tile" of buffer 0)
  set_ptrs(0, 0);
#pragma unroll
  for (int idx = 0; idx < KS + 4; ++idx) far_read(idx);
  // Scores of the first tile through TEMPORARY fragments: their maximum fixes the reference (see qfwd2_kernel)
  float ref_a;
  {
    bf16x8 tmpf[KS];
    {
      int64_t row = rblk * 128 + wave * 32 + r;
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
    const float rv = (m == NEG_INF_F) ? 0.f : m + QF3_REF_MARGIN;
    rl2 = -rv * CQL_LOG2E;
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ref_a) : "v"(rv));
    asm volatile("" : "+v"(rl2));
  }
  {
    int64_t row = rblk * 128 + wave * 32 + r;
    if (row >= a.n_states) row = a.n_states - 1;
#pragma unroll
    for (int s = 0; s < KS; ++s) rf[s] = *reinterpret_cast<const bf16x8*>(a.H_b + row * D + 16 * s + 8 * h);
    __builtin_amdgcn_s_waitcnt(0x0F70);       // (see qde2_kernel::load_owner)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  acc0 = sv;        // scores of tile 0
#pragma unroll
  for (int s = 0; s < KS; ++s) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], rf[s], acc0, 0, 0, 0);
  if (nst > 1) {    // rows + bias of tile 1
    set_ptrs(0, 1);
#pragma unroll
    for (int idx = 0; idx < KS + 4; ++idx) far_read(idx);
  }

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  auto rotate = [&]() __attribute__((always_inline)) {
    const int t_ = b_cur;
    b_cur = b_mid;
    b_mid = b_far;
    b_far = t_;
  };
  for (st = 0; st + 1 < nst; st += 2) {       // two tiles per trip: the accumulator parity is a compile-time constant
    set_ptrs(b_cur, b_far);
    period(I0{});
    rotate();
    ++st;
    set_ptrs(b_cur, b_far);
    period(I1{});
    rotate();
    --st;
  }
  if (st < nst) {
    set_ptrs(b_cur, b_far);
    period(I0{});
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last products have left the pipe before Y is read
  de_wait_vmcnt<0>();      // the stages issued past the slice have landed before this wave gives its LDS back

  // ---- partials: (reference, sum relative to it) and the un-normalised slab ---------------------------------------------
  {
    const int64_t row = rblk * 128 + wave * 32 + r;
    const float ls = cs + __shfl_xor(cs, 32);
    if (row < a.n_states) {
      const int64_t pidx = (int64_t)split * a.n_states + row;
      float* dst = a.slab + pidx * D;
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(dst + ft * 32 + 8 * q + 4 * h) =
              make_float4(y[ft][4 * q + 0], y[ft][4 * q + 1], y[ft][4 * q + 2], y[ft][4 * q + 3]);
      if (h == 0) {
        float rv;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(rv) : "a"(ref_a));
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
bool cql_qfwd3_supported(int d, int64_t n_items) {
  static const int off = getenv("CQL_QFWD3") && getenv("CQL_QFWD3")[0] == '0';
  return !off && d == 256 && n_items * 512 < (1ll << 31);
}

int cql_qfwd3_run(const QFwd2Args& a, int d, hipStream_t s) {
  if (!cql_qfwd3_supported(d, a.n_items)) return CQLREC_ERR_INVALID;
  constexpr int smem = QF3_NBUF * DeCfg<256, 4>::BUF_BYTES;
  const int64_t rblks = (a.n_states + 127) / 128;
  hipLaunchKernelGGL((qfwd3_kernel<256>), dim3((unsigned)(rblks * a.nsplit)), dim3(256), smem, s, a);
  CQL_LAUNCH_CHECK("qfwd3");
  return CQLREC_OK;
}
