// Item-side backward of the Q-head (row a6: g_E_out, g_b_out), the longest kernel of the training step, as a kernel of
// its own: a PERSISTENT, statically balanced ("stream-K") version of the BWD_DE mode of qhead.hip.
//
//   dE^T[f][item] = sum_state H_b^T[f][state] * bf16(P[state][item]),   P = exp2((S + b_item) log2e - lse_state log2e)
//   S[state][item] = <H_b[state], E_out_b[item]>        (recomputed: the B x N score matrix never exists)
//
// What is different from the generic streaming skeleton, and why (measured there: 0.255 ms at cfg3 = 35 % MFMA busy,
// VALU / MFMA instruction ratio 5.8, 782 blocks on 512 resident slots):
//  * work = G item groups (128 items: 32 per wave, owner fragments in registers) x T stages of 64 states.  The G*T
//    stage-units are cut into `gridDim.x` CONTIGUOUS, EQUAL ranges, one per persistent block (2 per CU), so every SIMD gets
//    the same number of MFMAs -- the generic kernel gives a SIMD 3 or 4 item tiles (24 % of the slot-time idle at
//    N = 100 000).  A group that straddles two ranges is summed in two pieces: the piece that holds stage 0 goes straight
//    to g_E_out (on top of the one-hot part already scattered there), the other one to a slab that a small fix-up
//    kernel adds afterwards, in block order -- deterministic, no atomics.  The stream of state tiles just wraps around at a
//    group boundary, so the LDS-DMA ring never drains there.
//  * zero VALU instructions for staging: `buffer_load_dwordx4 ... offen lds` with a per-lane constant voffset and the
//    stage offset in an SGPR (the generic kernel spends ~35 VALU per stage on 64-bit addresses and tail clamps; the
//    buffer descriptor's range check replaces the clamps);
//  * LDS image in 8-row x 32-column subtiles of 512 B (cdna_hip_programming.md T10, form (a)): the 8 row reads of a tile
//    use 2 per-lane base registers and the 16 transposed reads 2 more, everything else is an `offset:` immediate (the
//    256-byte-row image of the generic kernel needs 16 + 8 address registers and one v_add per transposed read).
// The arithmetic per element is unchanged (same MFMA chains, same exp2 / fmaf / bf16 rounding / summation order inside a
// piece), so a group that is not cut gives bit-identical rows to the generic kernel.
#include <stdlib.h>

#include "qhead_de_common.h"

template <int D, int NBUF, int WAVES>
__global__ __launch_bounds__(64 * WAVES, (D == 256 ? 1 : 2)) void qde_kernel(QDeArgs a) {
  using C = DeCfg<D, WAVES>;
  constexpr int QDE_ITEMS = C::ITEMS;
  constexpr int PD = NBUF - 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // the ONLY LDS object of this kernel

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  // ---- this block's range of stage-units ---------------------------------------------------------------------
  const int64_t W = (int64_t)a.G * a.T;
  const int64_t u0 = (int64_t)blockIdx.x * W / gridDim.x, u1 = ((int64_t)blockIdx.x + 1) * W / gridDim.x;
  const int nst = (int)(u1 - u0);
  if (nst <= 0) return;
  unsigned long long stamp_tk = 0, stamp_rt = 0;
  if (a.stamps) qde_stamp(stamp_tk, stamp_rt);
  int g = (int)(u0 / a.T);               // current item group
  int t = (int)(u0 - (int64_t)g * a.T);  // current stage inside the group
  int t_seg = t;                         // first stage of the current piece of the group
  int t_dma = t;                         // stage of the next DMA to issue

  // ---- staging geometry (per-lane constants) -------------------------------------------------------------------
  __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)a.H_b, 0, (int)(a.n_states * C::ROWB), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)a.nlse2, 0, (int)(a.n_states * 4), 0x00020000);
  // piece pc = 4 i + wave of a stage: 8-row group rg = pc / PPG, chunk octet hc = pc % PPG; lane l fills image bytes
  // [16 l, 16 l + 16) of the piece: subtile l >> 5, row (l >> 2) & 7, slot l & 3 = (chunk & 3) ^ ((row >> 2) & 3)
  uint32_t voff[2];
  {
    const int sub = lane >> 5, r7 = (lane >> 2) & 7, slot = lane & 3;
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      // piece WAVES i + wave: rg = (WAVES / PPG) i + wave / PPG.  Its parity is independent of i unless WAVES / PPG is
      // odd (d=256 with 4 waves), where it alternates with i (`par`)
      const int rg0 = wave / C::PPG, hc = wave % C::PPG;                     // piece i = 0
      const int rg1 = (((WAVES / C::PPG) & 1) ? (rg0 + par) : rg0) & 1;
      const int q2 = (r7 >> 2) | (rg1 << 1);
      voff[par] = (uint32_t)((rg0 * 8 + r7) * C::ROWB + (8 * hc + 4 * sub + (slot ^ q2)) * 16);
    }
  }
  const uint32_t voff_strip = (uint32_t)lane * 4;
  constexpr bool PAR_ALT = ((WAVES / C::PPG) & 1) != 0;
  const uint32_t smem_base = lds_addr_of(smem);
  auto issue = [&](int stage_t, int buf) {
    const uint32_t bufp = __builtin_amdgcn_readfirstlane(smem_base + buf * C::BUF_BYTES);
    const uint32_t soff = (uint32_t)stage_t * C::STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < C::LPS; ++i)
      bdma16(voff[PAR_ALT ? (i & 1) : 0], rs_h, soff + C::PSTEP * i, bufp + (WAVES * i + wave) * 1024);
    if (wave == (stage_t & (WAVES - 1)))
      bdma4(voff_strip, rs_s, (uint32_t)stage_t * (C::TI * 4), bufp + C::STAGE_BYTES);
  };

  // ---- read geometry -----------------------------------------------------------------------------------------
  const lds_u8* lbase = (const lds_u8*)smem;
  // A operand of the S chain: row r, chunk 2 s + h  ->  base[s & 1] + 512 (s >> 1)
  const lds_u8* pA[2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
    pA[e] = lbase + C::RG_BYTES * (r >> 3) + 64 * (r & 7) + 16 * ((2 * e + h) ^ ((r >> 2) & 3));
  // transposed reads: lane 4 q + p of 16-lane group (g1, h): state row 16 s2 + 8 jj + 4 h + q, features 32 ft + 16 g1 + 4 p ..
  const lds_u8* pT[2];
  {
    const int g1 = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
      pT[jj] = lbase + 64 * (4 * h + q) + 16 * ((2 * g1 + (p >> 1)) ^ ((2 * jj + h) & 3)) + 8 * (p & 1);
  }
  const lds_u8* pS = lbase + C::STAGE_BYTES + 16 * h;                  // strip: 4 floats at 128 it + 32 q + 16 h

  // ---- owner state ---------------------------------------------------------------------------------------------
  bf16x8 rf[C::KS];
  f32x16 cinit;
  f32x16 y[C::FT];
  float cs = 0.f;
  auto load_owner = [&](int grp) {
    int64_t row = (int64_t)grp * QDE_ITEMS + wave * 32 + r;
    if (row >= a.n_items) row = a.n_items - 1;
#pragma unroll
    for (int s = 0; s < C::KS; ++s) rf[s] = *reinterpret_cast<const bf16x8*>(a.E_b + row * D + 16 * s + 8 * h);
    float bv = a.bias[row];
    // these ordinary loads must be retired -- in hipcc's own bookkeeping -- before the next LDS-DMA is issued: its counted
    // waits assume that nothing younger than its loads is in flight (cdna_hip_programming.md 5, trap (b))
#pragma unroll
    for (int s = 0; s < C::KS; ++s) {
      u32x4 tt = __builtin_bit_cast(u32x4, rf[s]);
      asm volatile("" : "+v"(tt));
      rf[s] = __builtin_bit_cast(bf16x8, tt);
    }
    asm volatile("" : "+v"(bv));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) cinit[i] = bv;
#pragma unroll
    for (int ft = 0; ft < C::FT; ++ft)
#pragma unroll
      for (int i = 0; i < 16; ++i) y[ft][i] = 0.f;
    cs = 0.f;
  };
  auto store_piece = [&](int grp, bool first) {
    const int64_t row = (int64_t)grp * QDE_ITEMS + wave * 32 + r;
    const bool ok = row < a.n_items;
    const float csum = cs + __shfl_xor(cs, 32);
    if (first) {
      if (ok) {
        float* dst = a.out + row * D;
#pragma unroll
        for (int ft = 0; ft < C::FT; ++ft)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float4* pd = reinterpret_cast<float4*>(dst + ft * 32 + 8 * q + 4 * h);
            float4 o = make_float4(a.scale * y[ft][4 * q + 0], a.scale * y[ft][4 * q + 1], a.scale * y[ft][4 * q + 2],
                                   a.scale * y[ft][4 * q + 3]);
            if (a.accumulate) {
              const float4 old = *pd;
              o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
            }
            *pd = o;
          }
        if (h == 0) a.out_cs[row] = a.accumulate ? a.out_cs[row] + a.scale * csum : a.scale * csum;
      }
    } else {   // a piece that starts past stage 0: unscaled, to this block's slab (rows past n_items: harmless, not read)
      float* dst = a.slab + ((int64_t)blockIdx.x * QDE_ITEMS + wave * 32 + r) * D;
#pragma unroll
      for (int ft = 0; ft < C::FT; ++ft)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(dst + ft * 32 + 8 * q + 4 * h) =
              make_float4(y[ft][4 * q + 0], y[ft][4 * q + 1], y[ft][4 * q + 2], y[ft][4 * q + 3]);
      if (h == 0) a.slab_cs[(int64_t)blockIdx.x * QDE_ITEMS + wave * 32 + r] = csum;
    }
  };

  load_owner(g);

  // ---- prologue of the ring --------------------------------------------------------------------------------------
  int issued = 0;
#pragma unroll
  for (int s0 = 0; s0 < PD; ++s0)
    if (s0 < nst) {
      issue(t_dma, s0);
      ++issued;
      if (++t_dma == a.T) t_dma = 0;
    }

  // NBUF stages per trip of the outer loop, inner fully unrolled: every LDS address is "per-lane base + immediate"
  for (int j0 = 0; j0 < nst; j0 += NBUF) {
#pragma unroll
    for (int sb = 0; sb < NBUF; ++sb) {
      const int j = j0 + sb;
      if (j >= nst) break;
      // stage j has landed once at most the younger in-flight stages' pieces are outstanding
      // (a wave that also carried the strip of the younger stage has LPS + 1 younger pieces: vmcnt(LPS) over-waits by one)
      if (PD >= 2 && issued - j - 1 >= 1) de_wait_vmcnt<C::LPS>();
      else de_wait_vmcnt<0>();
#ifndef QDE_ABL_NOBAR
      __builtin_amdgcn_s_barrier();          // everyone's pieces landed; everyone left the buffer refilled next
#endif
#ifdef QDE_ABL_NODMA      // timing-only build: the ring is filled once, never refilled
      if (false) {
#else
      if (issued < nst) {
#endif
        issue(t_dma, (sb + PD) % NBUF);
        ++issued;
        if (++t_dma == a.T) t_dma = 0;
      }
      const int64_t state0 = (int64_t)t * C::TI;

#pragma unroll
      for (int it = 0; it < C::TILES; ++it) {
        const int64_t left = a.n_states - (state0 + 32 * it);          // states of this tile that exist (block-uniform)
        if (left <= 0) continue;
        const int toff = sb * C::BUF_BYTES + it * C::TILE_BYTES;
        f32x16 sv;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 t4 = *(const lds_f4*)(pS + sb * C::BUF_BYTES + 128 * it + 32 * q);
          sv[4 * q + 0] = t4[0];
          sv[4 * q + 1] = t4[1];
          sv[4 * q + 2] = t4[2];
          sv[4 * q + 3] = t4[3];
        }
        f32x16 acc = cinit;
        {
          bf16x8 af[C::KS];
#pragma unroll
          for (int s = 0; s < C::KS; ++s) af[s] = *(const lds_bf16x8*)(pA[s & 1] + toff + 512 * (s >> 1));
#pragma unroll
          for (int s = 0; s < C::KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], rf[s], acc, 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, C::KS + 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, C::KS, 0);
        }
        if (left < 32) {   // states past the end of the batch: probability 0
#pragma unroll
          for (int i = 0; i < 16; ++i) sv[i] = (mfma_row(i, h) < left) ? sv[i] : NEG_INF_F;
        }
        bf16x8 tf[C::FT][2];
#pragma unroll
        for (int ft = 0; ft < C::FT; ++ft)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              const bf16x4 t4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                  (lds_bf16x4*)(pT[jj] + toff + C::RG_BYTES * (2 * s2 + jj) + 512 * ft));
              tf[ft][s2][4 * jj + 0] = t4[0];
              tf[ft][s2][4 * jj + 1] = t4[1];
              tf[ft][s2][4 * jj + 2] = t4[2];
              tf[ft][s2][4 * jj + 3] = t4[3];
            }
        float p[16];
#pragma unroll
#if defined(QDE_ABL_NOVALU)   // timing-only: no exp / fma / column sums
        for (int i = 0; i < 16; ++i) p[i] = acc[i];
        asm volatile("" : "+v"(sv[0]));
#elif defined(QDE_ABL_NOEXP)  // timing-only: the fma stays, the transcendental goes
        for (int i = 0; i < 16; ++i) p[i] = fmaf(acc[i], CQL_LOG2E, sv[i]);
#else
        for (int i = 0; i < 16; ++i) p[i] = fast_exp2(fmaf(acc[i], CQL_LOG2E, sv[i]));
#endif
#ifndef QDE_ABL_NOVALU
        float c1 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) c1 += p[i];
        cs += c1;
#endif
        bf16x8 pf[2];
#pragma unroll
        for (int jx = 0; jx < 8; ++jx) {
          pf[0][jx] = (__bf16)p[jx];
          pf[1][jx] = (__bf16)p[8 + jx];
        }
#pragma unroll
        for (int ft = 0; ft < C::FT; ++ft)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) y[ft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[ft][s2], pf[s2], y[ft], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, C::FT * 4, 1);
        __builtin_amdgcn_sched_group_barrier(0x402, QDE_VALU_HINT, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, C::FT * 2, 1);
      }

      // ---- end of a stage: group / range boundaries -------------------------------------------------------------
      const bool last = (j == nst - 1);
      if (t == a.T - 1 || last) {
        store_piece(g, t_seg == 0);
        if (!last) {
          ++g;
          t = -1;
          t_seg = 0;
          load_owner(g);
        }
      }
      ++t;
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

// out[group rows] += scale * slab pieces of the blocks whose range starts inside the group (in block order)
template <int D, int QDE_ITEMS>
__global__ __launch_bounds__(256) void qde_fixup_kernel(QDeArgs a, int nblk) {
#pragma clang fp contract(off)      // product, then sum: the bits of the deferred fix-up (cql_adam_ema_fix, misc.hip)
  const int g = blockIdx.x;
  const int64_t W = (int64_t)a.G * a.T;
  const int64_t lo = (int64_t)g * a.T, hi = lo + a.T;
  // first block p with u0(p) = floor(p W / nblk) > lo
  int64_t p = ((lo + 1) * nblk + W - 1) / W;
  for (; p < nblk; ++p) {
    const int64_t u0 = p * W / nblk;
    if (u0 >= hi) break;
    if (u0 <= lo) continue;
    if (((p + 1) * W / nblk) == u0) continue;          // an empty range wrote nothing
    const float* src = a.slab + p * QDE_ITEMS * D;
    for (int idx = threadIdx.x; idx < QDE_ITEMS * (D / 4); idx += blockDim.x) {
      const int row = idx / (D / 4), c = idx % (D / 4);
      const int64_t item = (int64_t)g * QDE_ITEMS + row;
      if (item >= a.n_items) continue;
      const float4 s = *reinterpret_cast<const float4*>(src + (int64_t)row * D + c * 4);
      float4* pd = reinterpret_cast<float4*>(a.out + item * D + c * 4);
      float4 o = *pd;
      o.x += a.scale * s.x; o.y += a.scale * s.y; o.z += a.scale * s.z; o.w += a.scale * s.w;
      *pd = o;
      if (c == 0) a.out_cs[item] += a.scale * a.slab_cs[p * QDE_ITEMS + row];
    }
    __syncthreads();
  }
}

// -lse in natural units from -lse*log2e (what the ABI carries): the strip of qde2_kernel is the C operand of its S chains
__global__ void qde_nlse_natural_kernel(const float* __restrict__ nlse2, int64_t n, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = nlse2[i] * CQL_LN2;
}

// =============================================================================================================
// host side
// =============================================================================================================
static inline int64_t de_align256(int64_t x) { return (x + 255) / 256 * 256; }
static int de_env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}
// block shape: 8 waves (256 items per group, ONE block per CU, both waves of a SIMD stream the same tiles: half the LDS
// fill traffic and half the LDS-DMA instructions per MFMA of the 4-wave form) unless the register budget forbids two
// waves per SIMD (d = 256)
static int de_waves(int d) {
  static const int w = de_env_int("CQL_QDE_WAVES", 8);
  return (d == 256 || w != 8) ? 4 : 8;
}
// form 2 (64 items per wave, one wave per SIMD, in-wave pipeline: qhead_de2.hip) for d = 128 unless CQL_QDE2=0
static bool de_form2(int d, int64_t batch) {
  static const int on = de_env_int("CQL_QDE2", 1);
  return on != 0 && d == 128 && batch % 64 == 0;
}
static int de_grid(int64_t n_items, int64_t batch, int d) {
  const int ti = (d == 256) ? 32 : 64, items = de_form2(d, batch) ? 256 : 32 * de_waves(d);
  const int64_t G = (n_items + items - 1) / items, T = (batch + ti - 1) / ti;
  static const int per_cu = de_env_int("CQL_QDE_BLOCKS_PER_CU", 0);
  static const int n_cu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        n <= 0 || n > 256)
      n = 256;
    return n;
  }();
  int64_t slots = (int64_t)n_cu * (per_cu > 0 && !de_form2(d, batch) ? per_cu : (d == 256 || de_form2(d, batch) || de_waves(d) == 8 ? 1 : 2));
  const int64_t W = G * T;
  return (int)(W < slots ? W : slots);
}

int64_t cql_qde_ws_bytes(int64_t batch, int64_t n_items, int32_t d) {
  const int64_t grid = 512;      // upper bound of de_grid() for every block shape
  return de_align256(grid * QDE_MAX_ITEMS * d * 4) + de_align256(grid * QDE_MAX_ITEMS * 4) + de_align256(batch * 4) + de_align256(grid * 16) + 256;
}

template <int D, int NBUF, int WAVES>
static void qde_launch_n(const QDeArgs& a, int grid, hipStream_t s) {
  constexpr int smem = NBUF * DeCfg<D, WAVES>::BUF_BYTES;
  static bool attr_set_dev[CQL_MAX_DEVICES] = {};   // > 64 KiB of dynamic LDS needs the opt-in once per kernel and device
  bool& attr_set = attr_set_dev[cql_device_slot()];
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)qde_kernel<D, NBUF, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  hipLaunchKernelGGL((qde_kernel<D, NBUF, WAVES>), dim3(grid), dim3(64 * WAVES), smem, s, a);
}
template <int D>
static void qde_launch_d(const QDeArgs& a, int grid, int nbuf, int waves, hipStream_t s) {
  if constexpr (D != 256) {
    if (waves == 8) {
      if (nbuf == 3) qde_launch_n<D, 3, 8>(a, grid, s); else qde_launch_n<D, 2, 8>(a, grid, s);
      return;
    }
  }
  if (nbuf == 3) qde_launch_n<D, 3, 4>(a, grid, s); else qde_launch_n<D, 2, 4>(a, grid, s);
}
template <int D>
static void qde_fixup_d(const QDeArgs& a, int grid, int waves, hipStream_t s) {
  if (waves == 8) hipLaunchKernelGGL((qde_fixup_kernel<D, 256>), dim3(a.G), dim3(256), 0, s, a, grid);
  else hipLaunchKernelGGL((qde_fixup_kernel<D, 128>), dim3(a.G), dim3(256), 0, s, a, grid);
}

// g_E_out / g_b_out rows [0, n_items) of this call: out (+)= scale * dE (accumulate: on top of what is there)
int cql_qde_launch(const uint16_t* H_b, const float* nlse2, int64_t batch, const uint16_t* E_b, const float* bias,
                   int64_t n_items, int32_t d, float scale, void* ws, int64_t ws_bytes, float* out, float* out_cs,
                   int accumulate, hipStream_t s, CqlAdamFix* defer, const float* nlse_nat) {
  if (defer) defer->valid = 0;
  CQL_REQUIRE(ws_bytes >= cql_qde_ws_bytes(batch, n_items, d), "qde: workspace too small");
  CQL_REQUIRE(batch * 2 * d < (1ll << 31), "qde: batch=%lld too large for one buffer descriptor", (long long)batch);
  const bool form2 = de_form2(d, batch);
  const int ti = (d == 256) ? 32 : 64, waves = form2 ? 8 : de_waves(d), items = form2 ? 256 : 32 * waves;
  QDeArgs a = {};
  a.H_b = H_b;
  a.nlse2 = nlse2;
  a.n_states = batch;
  a.E_b = E_b;
  a.bias = bias;
  a.n_items = n_items;
  a.out = out;
  a.out_cs = out_cs;
  a.scale = scale;
  a.accumulate = accumulate;
  a.G = (int32_t)((n_items + items - 1) / items);
  a.T = (int32_t)((batch + ti - 1) / ti);
  const int grid = de_grid(n_items, batch, d);
  a.slab = (float*)ws;
  a.slab_cs = (float*)((char*)ws + de_align256((int64_t)grid * items * d * 4));
  static const int nbuf = de_env_int("CQL_QDE_NBUF", 2);
  static const int want_stamps = de_env_int("CQL_QDE_STAMPS", 0);     // diagnostic: in-kernel clock, synchronises
  unsigned long long* stamps_dev = (unsigned long long*)((char*)a.slab_cs + de_align256((int64_t)grid * items * 4) +
                                                         de_align256(batch * 4));
  a.stamps = want_stamps ? stamps_dev : nullptr;
  const bool form3 = !form2 && cql_qde3_supported(d, batch);      // d = 256: same groups, stages and grid as qde_kernel<256>
  if (form2 || form3) {
    if (nlse_nat) {        // the caller's forward already wrote -lse in natural units
      a.nlse2 = nlse_nat;
    } else {
      float* nat = (float*)((char*)a.slab_cs + de_align256((int64_t)grid * items * 4));
      {
        CqlProfScope prof(CQLREC_PH_QHEAD_SMALL, s);
        hipLaunchKernelGGL(qde_nlse_natural_kernel, dim3(cql_ceil_div(batch, 256)), dim3(256), 0, s, nlse2, batch, nat);
      }
      a.nlse2 = nat;
    }
    CqlProfScope prof(CQLREC_PH_QHEAD_BWD_DE, s);
    const int rc = form2 ? cql_qde2_run(a, d, grid, s) : cql_qde3_run(a, d, grid, s);
    if (rc != CQLREC_OK) return rc;
  } else {
    CqlProfScope prof(CQLREC_PH_QHEAD_BWD_DE, s);
    if (d == 64) qde_launch_d<64>(a, grid, nbuf, waves, s);
    else if (d == 128) qde_launch_d<128>(a, grid, nbuf, waves, s);
    else qde_launch_d<256>(a, grid, nbuf, waves, s);
  }
  const int64_t W = (int64_t)a.G * a.T;
  bool cut = false;                    // does some range start inside a group?
  for (int p = 1; p < grid && !cut; ++p) cut = ((int64_t)p * W / grid) % a.T != 0;
  if (cut && defer) {     // the consumer of the gradient adds the slabs (see CqlAdamFix): same terms, same order
    defer->valid = 1;
    defer->slab = a.slab;
    defer->slab_cs = a.slab_cs;
    defer->G = a.G;
    defer->T = a.T;
    defer->nblk = grid;
    defer->items = items;
    defer->D = d;
    defer->scale = a.scale;
    defer->n_items = n_items;
  } else if (cut) {
    CqlProfScope prof(CQLREC_PH_QHEAD_SMALL, s);
    if (d == 64) qde_fixup_d<64>(a, grid, waves, s);
    else if (d == 128) qde_fixup_d<128>(a, grid, waves, s);
    else qde_fixup_d<256>(a, grid, waves, s);
  }
  CQL_LAUNCH_CHECK("qde");
  if (want_stamps) {      // median over blocks of shader ticks / 100 MHz ticks (MI355X_MICROARCH.md, DVFS give-back item 6)
    static unsigned long long host[1024];
    if (hipStreamSynchronize(s) == hipSuccess &&
        hipMemcpy(host, stamps_dev, (size_t)grid * 16, hipMemcpyDeviceToHost) == hipSuccess) {
#ifdef QDE2_STAMP_TURN     // diagnostic build (tools/build_variant.sh x -DQDE2_STAMP_TURN): the second word is packed turn ticks
      double tot = 0, lg = 0, vm = 0, bar = 0;
      for (int i = 0; i < grid; ++i) {
        tot += (double)host[2 * i];
        lg += (double)((host[2 * i + 1] >> 40) & 0xFFFFF);
        vm += (double)((host[2 * i + 1] >> 20) & 0xFFFFF);
        bar += (double)(host[2 * i + 1] & 0xFFFFF);
      }
      fprintf(stderr, "[qde2 turn stamps] blocks=%d mean ticks/block %.0f; parked at ring turns (wave 0, raw, incl. ~2 stamp "
              "latencies each): lgkmcnt %.1f %%  vmcnt %.1f %%  barrier %.1f %%\n", grid, tot / grid, 100 * lg / tot,
              100 * vm / tot, 100 * bar / tot);
      return CQLREC_OK;
#endif
      double best_clk[512];
      double rt_max = 0;
      for (int i = 0; i < grid; ++i) {
        best_clk[i] = host[2 * i + 1] ? (double)host[2 * i] / (double)host[2 * i + 1] * 0.1 : 0.0;
        if ((double)host[2 * i + 1] > rt_max) rt_max = (double)host[2 * i + 1];
      }
      for (int i = 0; i < grid; ++i)
        for (int j = i + 1; j < grid; ++j)
          if (best_clk[j] < best_clk[i]) { double x = best_clk[i]; best_clk[i] = best_clk[j]; best_clk[j] = x; }
      fprintf(stderr, "[qde stamps] blocks=%d in-kernel clock median %.3f GHz (min %.3f max %.3f), longest block %.1f us\n", grid,
              best_clk[grid / 2], best_clk[0], best_clk[grid - 1], rt_max * 0.01);
    }
  }
  return CQLREC_OK;
}

// the fix-up launch for a descriptor cql_qde_launch filled instead of launching it (a caller that has something to add to
// the rows between the long kernel and the slabs: the one-hot part, see train.hip)
int cql_qde_fixup_deferred(const CqlAdamFix& f, float* out, float* out_cs, hipStream_t s) {
  if (!f.valid) return CQLREC_OK;
  QDeArgs a = {};
  a.n_items = f.n_items;
  a.out = out;
  a.out_cs = out_cs;
  a.scale = f.scale;
  a.slab = const_cast<float*>(f.slab);
  a.slab_cs = const_cast<float*>(f.slab_cs);
  a.G = f.G;
  a.T = f.T;
  CqlProfScope prof(CQLREC_PH_QHEAD_SMALL, s);
  const int waves = f.items / 32;
  if (f.D == 64) qde_fixup_d<64>(a, f.nblk, waves, s);
  else if (f.D == 128) qde_fixup_d<128>(a, f.nblk, waves, s);
  else qde_fixup_d<256>(a, f.nblk, waves, s);
  CQL_LAUNCH_CHECK("qde fix-up");
  return CQLREC_OK;
}
