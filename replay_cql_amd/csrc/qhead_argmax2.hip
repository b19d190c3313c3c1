// ARGMAX pass of the training step (a* = argmax_j Q(s', j), SURVEY 8.0 S5: the action of the double-Q target) as a
// one-wave-per-SIMD, software-pipelined kernel for d = 128 and d = 256 -- the score-only sibling of qfwd2 / qfwd3 / qtopk2.
// Role in the reference's structure: the forward over the whole catalogue of the torch analogues
// (replay/models/mult_vae.py:101); no reference counterpart of the arg-max itself.
//
// A wave owns TWO 32-state groups (256 states per block); the items stream through a 2-deep LDS ring (LDS-DMA, the image
// of qde_kernel).  Per 32-item tile two chains of KS MFMAs (C operand of the first product = the items' bias), and behind
// each chain its epilogue in the gaps of the NEXT chain: the maximum of the lane's 16 scores (v_max3 tree) and a
// branch-free "first tile in which the running maximum was reached" update -- what QM_ARGMAX of qstream_kernel keeps.
// The rows and bias of tile t+1 are read, one LDS read per gap, while the chains of tile t run (second register set).
// Output per (item slice, state): (maximum, first row of its tile); qhead_argmax_resolve_kernel (qhead.hip) picks the
// slice and finds the position inside the tile by re-running that tile's chain -- the SAME chain as here (bias first, then
// the d / 16 products in order), so the scores it compares are bit-identical.
//
// There is ONE form of the stage loop and no branch in it: behind the slice's last stage the ring keeps turning (the
// stages it issues lie past the slice; the buffer descriptor bounds them and nobody uses them).
#include <stdlib.h>
#include <type_traits>
#include "qhead_de_common.h"

struct QArgmax2Args {
  const uint16_t* H_b;      // [n_states x D] owner rows
  int64_t n_states;
  const uint16_t* E_b;      // [n_items x D] streamed rows
  const float* bias;        // [n_items]
  int64_t n_items;
  int nsplit;
  int64_t split_rows;       // items per slice (multiple of 64)
  float* part_v;            // [nsplit][n_states] maximum
  int32_t* part_i;          // [nsplit][n_states] first row of the tile it was reached in
};

__device__ __forceinline__ float am2_max3(float a, float b, float c) {
  float q;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(q) : "v"(a), "v"(b), "v"(c));
  return q;
}

template <int D>
__global__ __launch_bounds__(256, 1) void qargmax2_kernel(QArgmax2Args a) {
  using C = DeCfg<D, 4>;
  constexpr int KS = C::KS, TILES = C::TILES;
  constexpr bool PAR_ALT = ((4 / C::PPG) & 1) != 0;          // d = 256: the row group's parity alternates with the piece
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

  // ---- staging (see qde_kernel) -------------------------------------------------------------------------------------------
  __amdgpu_buffer_rsrc_t rs_e = __builtin_amdgcn_make_buffer_rsrc((void*)a.E_b, 0, (int)(a.n_items * C::ROWB), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, (int)(a.n_items * 4), 0x00020000);
  uint32_t voff[2];
  {
    const int sub = lane >> 5, r7 = (lane >> 2) & 7, slot = lane & 3;
    const int rg0 = wave / C::PPG, hc = wave % C::PPG;
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int rg1 = (PAR_ALT ? (rg0 + par) : rg0) & 1;
      const int q2 = (r7 >> 2) | (rg1 << 1);
      voff[par] = (uint32_t)((rg0 * 8 + r7) * C::ROWB + (8 * hc + 4 * sub + (slot ^ q2)) * 16);
    }
  }
  const uint32_t voff_strip = (uint32_t)lane * 4;
  const uint32_t smem_base = lds_addr_of(smem);
  auto issue_piece = [&](int stage, int buf, int i) __attribute__((always_inline)) {
    const uint32_t bufp = __builtin_amdgcn_readfirstlane(smem_base + buf * C::BUF_BYTES);
    const uint32_t gs = gst0 + (uint32_t)stage;
    bdma16(voff[PAR_ALT ? (i & 1) : 0], rs_e, gs * C::STAGE_BYTES + C::PSTEP * i, bufp + (4 * i + wave) * 1024);
  };
  auto issue_strip = [&](int stage, int buf) __attribute__((always_inline)) {
    const uint32_t bufp = __builtin_amdgcn_readfirstlane(smem_base + buf * C::BUF_BYTES);
    const uint32_t gs = gst0 + (uint32_t)stage;
    if (wave == (stage & 3)) bdma4(voff_strip, rs_b, gs * (C::TI * 4), bufp + C::STAGE_BYTES);
  };
  // items past the end of the catalogue (last stage of the last slice): bias -inf, so that they never are a maximum
  auto patch_strip = [&](int stage, int buf) __attribute__((always_inline)) {
    const int64_t valid = a.n_items - (int64_t)(gst0 + (uint32_t)stage) * C::TI;
    if (valid < C::TI) {
      if (wave == 0 && lane < C::TI && lane >= valid)
        *(__attribute__((address_space(3))) float*)((lds_u8*)smem + buf * C::BUF_BYTES + C::STAGE_BYTES + lane * 4) = NEG_INF_F;
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_s_barrier();
    }
  };

  // ---- read geometry ------------------------------------------------------------------------------------------------------
  const lds_u8* lbase = (const lds_u8*)smem;
  const int oa0 = C::RG_BYTES * (r >> 3) + 64 * (r & 7) + 16 * ((0 + h) ^ ((r >> 2) & 3));
  const int oa1 = C::RG_BYTES * (r >> 3) + 64 * (r & 7) + 16 * ((2 + h) ^ ((r >> 2) & 3));
  const int os = C::STAGE_BYTES + 16 * h;
  const lds_u8 *pA0, *pA1, *pS;        // current stage's buffer
  const lds_u8 *nA0, *nA1, *nS;        // next stage's buffer
  auto set_ptrs = [&](int bc, int bn) __attribute__((always_inline)) {
    pA0 = lbase + bc * C::BUF_BYTES + oa0; pA1 = lbase + bc * C::BUF_BYTES + oa1; pS = lbase + bc * C::BUF_BYTES + os;
    nA0 = lbase + bn * C::BUF_BYTES + oa0; nA1 = lbase + bn * C::BUF_BYTES + oa1; nS = lbase + bn * C::BUF_BYTES + os;
  };

  // ---- owner state ------------------------------------------------------------------------------------------------------
  bf16x8 rf[2][KS];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    int64_t row = rblk * 256 + wave * 64 + g * 32 + r;
    if (row >= a.n_states) row = a.n_states - 1;
#pragma unroll
    for (int s = 0; s < KS; ++s) rf[g][s] = *reinterpret_cast<const bf16x8*>(a.H_b + row * D + 16 * s + 8 * h);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);       // (see qde2_kernel::load_owner)
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float best[2] = {NEG_INF_F, NEG_INF_F};
  int btile[2] = {0x7FFFFFFF, 0x7FFFFFFF};

#define AM2_FENCE() __builtin_amdgcn_sched_barrier(0)
  bf16x8 af[2][KS];        // item-row fragments by tile parity
  f32x16 sv[2];            // bias of the tile (C operand of both chains)
  f32x16 acc0, acc1;
  float qm[3], tmx = 0.f;

  // LDS reads of the tile FOLLOWING tile IT of the current stage: idx 0..KS-1 rows, KS..KS+3 bias quarters
  auto next_read = [&](auto IT, auto P_, int idx) __attribute__((always_inline)) {
    constexpr bool END = decltype(IT)::value == TILES - 1;
    constexpr int NIT = END ? 0 : decltype(IT)::value + 1;
    constexpr int NP = decltype(P_)::value ^ 1;           // the other register set
    constexpr int noff = NIT * C::TILE_BYTES;
    if (idx < KS) {
      af[NP][idx] = *(const lds_bf16x8*)(((idx & 1) ? (END ? nA1 : pA1) : (END ? nA0 : pA0)) + noff + 512 * (idx >> 1));
    } else {
      const int q = idx - KS;
      const f32x4 t4 = *(const lds_f4*)((END ? nS : pS) + 128 * NIT + 32 * q);
      sv[NP][4 * q + 0] = t4[0];
      sv[NP][4 * q + 1] = t4[1];
      sv[NP][4 * q + 2] = t4[2];
      sv[NP][4 * q + 3] = t4[3];
    }
  };
  // epilogue of a finished chain, spread over gaps 3..6 of the chain that follows it
  auto epi = [&](int gp, const f32x16& acc, int g, int row0) __attribute__((always_inline)) {
    if (gp == 3) {
      qm[0] = am2_max3(acc[0], acc[1], acc[2]);
      qm[1] = am2_max3(acc[3], acc[4], acc[5]);
      qm[2] = am2_max3(acc[6], acc[7], acc[8]);
    } else if (gp == 4) {
      tmx = am2_max3(acc[9], acc[10], acc[11]);
      qm[0] = am2_max3(qm[0], qm[1], qm[2]);
    } else if (gp == 5) {
      tmx = am2_max3(tmx, acc[12], acc[13]);
      tmx = am2_max3(tmx, acc[14], acc[15]);
    } else if (gp == 6) {
      tmx = fmaxf(tmx, qm[0]);
      const bool upd = tmx > best[g];          // strictly: the FIRST tile keeps a tie
      best[g] = upd ? tmx : best[g];
      btile[g] = upd ? row0 : btile[g];
    }
  };

  int st = 0, cur_buf = 0;
  int issued = 0;
  bool turned = false;
  (void)turned;
  // the ring turns at the start of a stage's last tile: every wave has the stage's rows in registers (the last tile's were
  // read during the tile before), the next stage has landed (issued one stage ago); this stage's buffer takes stage + 2
  auto ring_turn = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    de_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    patch_strip(st + 1, cur_buf ^ 1);
  };

  // one tile: chain A (group 0) with the epilogue of the PREVIOUS tile's chain B, chain B (group 1) with the epilogue of A
  auto tile = [&](auto IT, auto P_, int row0, int row0_prev) __attribute__((always_inline)) {
    constexpr int P = decltype(P_)::value;
    constexpr bool END = decltype(IT)::value == TILES - 1;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (END && s == 0) ring_turn();
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[P][s], rf[0][s], s == 0 ? sv[P] : acc0, 0, 0, 0);
      AM2_FENCE();
      epi(s, acc1, 1, row0_prev);
      AM2_FENCE();
      if (END && s >= 1 && s <= C::LPS) issue_piece(issued, cur_buf, s - 1);      // refill, one piece per gap
      if (END && s == C::LPS + 1) issue_strip(issued, cur_buf);
      AM2_FENCE();
      next_read(IT, P_, s);
      AM2_FENCE();
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[P][s], rf[1][s], s == 0 ? sv[P] : acc1, 0, 0, 0);
      AM2_FENCE();
      epi(s, acc0, 0, row0);
      AM2_FENCE();
      if (s < 4) next_read(IT, P_, KS + s);
      AM2_FENCE();
    }
    if constexpr (END) ++issued;
  };

  // ---- prologue: two stages in flight; rows and bias of the first tile in registers ------------------------------------------
  for (int s0 = 0; s0 < 2; ++s0) {
#pragma unroll
    for (int i = 0; i < C::LPS; ++i) issue_piece(s0, s0, i);
    issue_strip(s0, s0);
    ++issued;
  }
  de_wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  patch_strip(0, 0);
  set_ptrs(1, 0);          // "the tile following the last tile of the stage before stage 0": next_read with buffer 0 as NEXT
#pragma unroll
  for (int idx = 0; idx < KS + 4; ++idx)
    next_read(std::integral_constant<int, TILES - 1>{}, std::integral_constant<int, 1>{}, idx);      // into set 0
#pragma unroll
  for (int i = 0; i < 16; ++i) acc1[i] = NEG_INF_F;      // "the chain B before the first tile": never a maximum

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  int row_prev = (int)s_begin;
  if constexpr (TILES == 2) {
    for (st = 0; st < nst; ++st) {
      set_ptrs(cur_buf, cur_buf ^ 1);
      const int row0 = (int)s_begin + st * C::TI;
      tile(I0{}, I0{}, row0, row_prev);
      tile(I1{}, I1{}, row0 + 32, row0);
      row_prev = row0 + 32;
      cur_buf ^= 1;
    }
  } else {          // one tile per stage: the register-set parity alternates with the stage -- two stages per trip
    for (st = 0; st < nst; ++st) {
      set_ptrs(cur_buf, cur_buf ^ 1);
      const int row0 = (int)s_begin + st * C::TI;
      tile(I0{}, I0{}, row0, row_prev);
      row_prev = row0;
      cur_buf ^= 1;
      if (++st >= nst) break;
      set_ptrs(cur_buf, cur_buf ^ 1);
      const int row1 = (int)s_begin + st * C::TI;
      tile(I0{}, I1{}, row1, row_prev);
      row_prev = row1;
      cur_buf ^= 1;
    }
  }
  // the last chain B's epilogue
#pragma unroll
  for (int gp = 3; gp <= 6; ++gp) epi(gp, acc1, 1, row_prev);
  de_wait_vmcnt<0>();      // the stages issued past the slice have landed before this wave gives its LDS back

  // ---- partials: the two lanes of a state hold different rows of every tile -------------------------------------------------
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int64_t row = rblk * 256 + wave * 64 + g * 32 + r;
    const float v2 = __shfl_xor(best[g], 32);
    const int i2 = __shfl_xor(btile[g], 32);
    const bool take2 = (v2 > best[g]) || (v2 == best[g] && i2 < btile[g]);
    if (row < a.n_states && h == 0) {
      const int64_t pidx = (int64_t)split * a.n_states + row;
      a.part_v[pidx] = take2 ? v2 : best[g];
      a.part_i[pidx] = take2 ? i2 : btile[g];
    }
  }
}

// =============================================================================================================
// host side
// =============================================================================================================
bool cql_qargmax2_supported(int d, int64_t n_items) {
  static const int off = getenv("CQL_QARGMAX2") && getenv("CQL_QARGMAX2")[0] == '0';
  return !off && (d == 128 || d == 256) && n_items * (2 * d) < (1ll << 31);
}

// slices for ONE block per CU (256 states per block); at least eight stages per slice
void cql_qargmax2_split(int64_t rows, int64_t n_items, int d, int* nsplit, int64_t* split_rows) {
  const int64_t rblks = (rows + 255) / 256;
  const int64_t units = (n_items + 63) / 64;
  int64_t want = (256 + rblks - 1) / rblks;
  if (want > units / 8) want = units / 8;
  if (want > 8) want = want / 8 * 8;          // whole multiples of the 8 XCDs: the row-blocks of a slice share an L2
  if (want < 1) want = 1;
  const int64_t upb = (units + want - 1) / want;
  *split_rows = upb * 64;
  *nsplit = (int)((n_items + *split_rows - 1) / *split_rows);
}

int cql_qargmax2_run(const uint16_t* H_b, int64_t rows, const uint16_t* E_b, const float* bias, int64_t n_items, int d,
                     int nsplit, int64_t split_rows, float* part_v, int32_t* part_i, hipStream_t s) {
  if (!cql_qargmax2_supported(d, n_items)) return CQLREC_ERR_INVALID;
  QArgmax2Args a = {H_b, rows, E_b, bias, n_items, nsplit, split_rows, part_v, part_i};
  const int64_t rblks = (rows + 255) / 256;
  if (d == 128) {
    constexpr int smem = 2 * DeCfg<128, 4>::BUF_BYTES;
    hipLaunchKernelGGL((qargmax2_kernel<128>), dim3((unsigned)(rblks * nsplit)), dim3(256), smem, s, a);
  } else {
    constexpr int smem = 2 * DeCfg<256, 4>::BUF_BYTES;
    hipLaunchKernelGGL((qargmax2_kernel<256>), dim3((unsigned)(rblks * nsplit)), dim3(256), smem, s, a);
  }
  CQL_LAUNCH_CHECK("qargmax2");
  return CQLREC_OK;
}
