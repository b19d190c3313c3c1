// Full-catalog Q-head on bf16 MFMA (v_mfma_f32_32x32x16_bf16), fused with its reductions so that the
// rows x N score matrix never reaches HBM.  One streaming skeleton serves five modes:
//
//   owner entity  = rows whose fragments stay in registers for the whole kernel (MFMA B operand, on the lane)
//   streamed entity = rows that flow HBM -> LDS (XOR-swizzled, double-buffered) and are the MFMA A operand
//
//   LSE / ARGMAX / TILEMAX : owner = states, streamed = items.   S[item][state] = E_out_b[item] . H_b[state] + b[item]
//        the per-state reduction over items is in-lane (16 registers) + one lane^32 exchange at the end.
//   BWD_DH : owner = states, streamed = items.   P = exp2(S*log2e - lse*log2e);  dH^T[f][state] += E_out_b^T P
//   BWD_DE : owner = items,  streamed = states.  P likewise;                     dE^T[f][item]  += H_b^T P
//        In both backward modes the 32x32 accumulator of S is converted to bf16 in registers and fed straight back
//        as the B operand of the second MFMA (k-order = accumulator row order); the A operand is the transposed
//        read (ds_read_b64_tr_b16) of the very tile already in LDS -- no second copy, no P in LDS or HBM.
//
// The streamed range is cut into `nsplit` slices; slice partials are merged by tiny finalize/reduce kernels
// (deterministic order).  blockIdx % nsplit = slice, so the blocks of one XCD (blockIdx % 8) share few slices.
#include <stdlib.h>
#include "qhead_internal.h"

static int qs_env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

template <int D>
struct QCfg {
  static constexpr int CPR = D / 8;                      // 16-byte chunks per row
  static constexpr int ROWB = D * 2;                     // bytes per row
  static constexpr int KS = D / 16;                      // MFMA k-steps per dot product
  static constexpr int TI = (D == 256) ? 32 : QS_TI;    // streamed rows per stage
  static constexpr int STAGE_BYTES = TI * ROWB;
  static constexpr int LPS = STAGE_BYTES / 1024 / 4;     // LDS-DMA instructions per wave per stage (1 KiB each)
  static constexpr int RPI = 64 / CPR;                   // rows covered by one LDS-DMA wave-instruction
  static constexpr int SC_BYTES = 4 * 64 * 4;            // per stage: one private 64-float scalar strip per wave
  static constexpr int BUF_BYTES = STAGE_BYTES + SC_BYTES;
};

#define NEG_INF (-__builtin_inff())
// VALU instructions the scheduler is asked to place between the transposed reads and the second MFMA chain
#ifndef QS_FUSED_VALU
#define QS_FUSED_VALU 64
#endif
#ifndef QS_BWD_VALU
#define QS_BWD_VALU 32
#endif
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(3))) f32x4 lds_f4;
typedef __attribute__((address_space(3))) bf16x8 lds_bf16x8;

// LDS image of a streamed tile: row r, 16-byte chunk ch lives at r*ROWB + (ch ^ f(r))*16.  f is chosen so that
//  * the MFMA A-operand row read (ds_read_b128, 16-lane groups with distinct rows mod 16, equal chunk) and
//  * the transposed read (ds_read_b64_tr_b16, per 32-lane half: 4 consecutive rows x 4 consecutive chunks)
// are both bank-conflict free (cdna_hip_programming.md T10 "one image for row reads AND transposed reads", form (b)).
template <int D>
__device__ __forceinline__ int swz2(int row, int ch) {
  if constexpr (D == 64) return ch ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
  else return ch ^ (((row & 3) << 2) | ((row >> 2) & 3));
}

// LDS-DMA through inline asm: hipcc models the builtin as an LDS store and would drain it (s_waitcnt vmcnt(0)) in
// front of the next ds_read; an asm statement is invisible to that bookkeeping, so the DMA queue is ours to count
// (cdna_hip_programming.md 5.7).  M0 carries the wave-uniform LDS byte address and is restored in the same statement.
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(lds_void_t*)p;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
  else static_assert(N < 0, "add the vmcnt literal");
}

// NBUF LDS stage buffers, filled by LDS-DMA (global_load_lds, no VGPR staging) NBUF-1 stages ahead of the MFMAs;
// one raw s_barrier per stage, counted vmcnt (never a drain inside the loop).
template <int D, int SPW, int MODE, int NBUF, int MINW>
__global__ __launch_bounds__(256, MINW) void qstream_kernel(QArgs a) {
  using C = QCfg<D>;
  constexpr bool FUSED = (MODE == QM_LSE_DH);          // forward LSE + softmax-weighted sum, running reference
  constexpr bool BWD = (MODE == QM_BWD_DH || MODE == QM_BWD_DE || FUSED);   // modes with the second MFMA
  constexpr int FT = D / 32;
  constexpr int PD = NBUF - 1;               // prefetch distance in stages
  constexpr int VPS = C::LPS + 1;            // vmcnt units per stage per wave (tile pieces + scalar strip)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // the ONLY LDS object of this kernel

  if constexpr (MODE == QM_LSE_DH) {     // fall-back launch behind qfwd2_kernel: runs only when that kernel asked for it
    if (a.guard && __builtin_nontemporal_load(a.guard) == 0) return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int split = blockIdx.x % a.nsplit;
  const int64_t rblk = blockIdx.x / a.nsplit;
  const int64_t res0 = (rblk * 4 + wave) * (32 * SPW);
  const int64_t s_begin = (int64_t)split * a.split_rows;
  const int64_t s_end = (s_begin + a.split_rows < a.n_str) ? (s_begin + a.split_rows) : a.n_str;
  const int nstage = (s_end > s_begin) ? (int)((s_end - s_begin + C::TI - 1) / C::TI) : 0;

  // ---- resident fragments ---------------------------------------------------------------------------------
  bf16x8 rf[SPW][C::KS];
  float rs[SPW];  // per-owner scalar (BWD_DH: -lse*log2e of the state; BWD_DE: bias of the item)
#pragma unroll
  for (int g = 0; g < SPW; ++g) {
    int64_t row = res0 + g * 32 + r;
    if (row >= a.n_res) row = a.n_res - 1;
#pragma unroll
    for (int s = 0; s < C::KS; ++s)
      rf[g][s] = *reinterpret_cast<const bf16x8*>(a.res + row * D + 16 * s + 8 * h);
    rs[g] = (BWD && !FUSED) ? a.res_scalar[row] : 0.f;
  }
  // The prologue's ordinary loads must be retired -- in hipcc's own bookkeeping -- before the loop: left alone, it
  // places their counted waits (vmcnt(7) ... vmcnt(0)) at the first use INSIDE the loop, and since our LDS-DMA pieces
  // share that counter every tile would drain the whole prefetch ring.  An empty asm that names each register forces
  // the wait here.
#pragma unroll
  for (int g = 0; g < SPW; ++g) {
#pragma unroll
    for (int s = 0; s < C::KS; ++s) {
      u32x4 t = __builtin_bit_cast(u32x4, rf[g][s]);
      asm volatile("" : "+v"(t));
      rf[g][s] = __builtin_bit_cast(bf16x8, t);
    }
    asm volatile("" : "+v"(rs[g]));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- per-mode state -------------------------------------------------------------------------------------
  float st_a[SPW], st_b[SPW];
  int st_i[SPW];
#pragma unroll
  for (int g = 0; g < SPW; ++g) {
    st_a[g] = NEG_INF;      // running max (LSE, ARGMAX, TILEMAX group max)
    st_b[g] = 0.f;          // running sum (LSE) / column sum of P (BWD_DE)
    st_i[g] = 0x7FFFFFFF;   // argmax
  }
  // QM_TOPK: per 32-user group the list of the QS_TOPK_K best admissible candidates of THIS lane's half of every tile
  // (sorted, best first; 0 = none), the pruning bound (the k-th best admissible score known for the user, shared by
  // the two lane halves), and a small LDS buffer of candidates that beat the bound since the last merge.
  constexpr bool TOPK = (MODE == QM_TOPK || MODE == QM_TOPK10);
  constexpr int TKC = (MODE == QM_TOPK10) ? 10 : QS_TOPK_K;      // list entries kept sorted
  unsigned long long tk_lst[TOPK ? SPW : 1][TOPK ? QS_TOPK_K : 1];
  float tk_thr[TOPK ? SPW : 1];
  int tk_cnt[TOPK ? SPW : 1];
  const uint32_t* tk_bits[TOPK ? SPW : 1];
  unsigned long long* tk_buf = nullptr;      // [SPW][QS_TOPK_BUF][64 lanes] of this wave
  if constexpr (TOPK) {
    tk_buf = reinterpret_cast<unsigned long long*>(smem + NBUF * C::BUF_BYTES) + (size_t)wave * SPW * QS_TOPK_BUF * 64;
#pragma unroll
    for (int g = 0; g < SPW; ++g) {
#pragma unroll
      for (int j = 0; j < QS_TOPK_K; ++j) tk_lst[g][j] = 0ull;
      tk_thr[g] = -3.0e38f;     // finite: rows past the end of the catalogue score -inf and never qualify
      tk_cnt[g] = 0;
      int64_t urow = res0 + g * 32 + r;
      if (urow >= a.n_res) urow = a.n_res - 1;
      tk_bits[g] = a.seen_bits ? a.seen_bits + urow * a.seen_w : nullptr;
    }
  }
  // Merge the buffered candidates of group g into its list (whole wave).  Seen items are dropped HERE: only candidates
  // that beat the bound are ever looked up (about k ln(n / k) per list), one bitmap word each, the loads of a merge in
  // flight together.  (hipcc waits vmcnt(0) for them, which also drains the prefetch ring -- once per merge.)
  auto tk_merge = [&](int g) {
    if constexpr (TOPK) {
      unsigned long long key[QS_TOPK_BUF];
      uint32_t wd[QS_TOPK_BUF];
#pragma unroll
      for (int e = 0; e < QS_TOPK_BUF; ++e)
        key[e] = (e < tk_cnt[g]) ? tk_buf[(g * QS_TOPK_BUF + e) * 64 + lane] : 0ull;
#pragma unroll
      for (int e = 0; e < QS_TOPK_BUF; ++e) {
        const uint32_t c = ~(uint32_t)(key[e] & 0xFFFFFFFFull);
        wd[e] = (key[e] != 0ull && tk_bits[g]) ? tk_bits[g][c >> 5] : 0u;
      }
#pragma unroll
      for (int e = 0; e < QS_TOPK_BUF; ++e) {
        const uint32_t c = ~(uint32_t)(key[e] & 0xFFFFFFFFull);
        unsigned long long kx = ((wd[e] >> (c & 31)) & 1u) ? 0ull : key[e];
#pragma unroll
        for (int j = 0; j < TKC; ++j) {       // insertion into the sorted list: keys are distinct
          const bool gt = kx > tk_lst[g][j];
          const unsigned long long hi_ = gt ? kx : tk_lst[g][j];
          kx = gt ? tk_lst[g][j] : kx;
          tk_lst[g][j] = hi_;
        }
      }
      tk_cnt[g] = 0;
      unsigned long long kk = 0ull;      // k-th entry through a static-index select chain (k is a run-time value)
#pragma unroll
      for (int j = 0; j < QS_TOPK_K; ++j) kk = (j == a.topk_k - 1) ? tk_lst[g][j] : kk;
      const float own = (kk != 0ull) ? f32_from_order_key((uint32_t)(kk >> 32)) : -3.0e38f;
      // k admissible items of the OTHER half at or above its bound exclude everything below that bound as well
      tk_thr[g] = fmaxf(own, __shfl_xor(own, 32));
    }
  };
  f32x16 y[BWD ? SPW : 1][BWD ? FT : 1];
  if constexpr (BWD) {
#pragma unroll
    for (int g = 0; g < SPW; ++g)
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int i = 0; i < 16; ++i) y[g][ft][i] = 0.f;
  }

  // ---- LDS-DMA staging: piece (wave, i) fills LDS bytes [(i*4 + wave) KiB, +1 KiB) of the stage buffer -------
  const int dma_row = lane / C::CPR, dma_chp = lane % C::CPR;   // row within the piece, LDS chunk slot
  const uint32_t smem_base = lds_addr(smem);
  auto issue = [&](int stage, int buf) {
    const uint32_t bufp = __builtin_amdgcn_readfirstlane(smem_base + buf * C::BUF_BYTES);
    const int64_t t0 = s_begin + (int64_t)stage * C::TI;
#pragma unroll
    for (int i = 0; i < C::LPS; ++i) {
      const int piece = i * 4 + wave;
      const int row = piece * C::RPI + dma_row;
      int64_t srow = t0 + row;
      if (srow >= a.n_str) srow = a.n_str - 1;
      const uint16_t* src = a.str + srow * D + swz2<D>(row, dma_chp) * 8;
      glds16(src, __builtin_amdgcn_readfirstlane(bufp + piece * 1024));
    }
    int64_t srow = t0 + lane;
    if (srow >= a.n_str) srow = a.n_str - 1;
    glds4(a.str_scalar + srow, __builtin_amdgcn_readfirstlane(bufp + C::STAGE_BYTES + wave * 256));
  };

  // transposed-read lane geometry (T10): lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3
  const int tr_fsub = ((lane >> 4) & 1) * 16, tr_q = (lane & 15) >> 2, tr_p = lane & 3;

#pragma unroll
  for (int s0 = 0; s0 < PD; ++s0)
    if (s0 < nstage) issue(s0, s0);

  // the ring position is a compile-time constant inside the body (NBUF stages per trip of the outer loop), so that
  // every LDS address of a tile is "per-lane constant + immediate offset": no address arithmetic per read
  for (int stage0 = 0; stage0 < nstage; stage0 += NBUF) {
#pragma unroll
   for (int sb = 0; sb < NBUF; ++sb) {
    const int stage = stage0 + sb;
    if (stage >= nstage) break;
    // stage `stage` has landed once at most the younger in-flight stages' pieces are outstanding
    const int younger = (nstage - 1 - stage < PD - 1) ? (nstage - 1 - stage) : (PD - 1);
    if (PD >= 3 && younger == 2) wait_vmcnt<2 * VPS>();
    else if (PD >= 2 && younger == 1) wait_vmcnt<VPS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // everyone's pieces of this stage landed; everyone left stage-1's buffer
    if (stage + PD < nstage) issue(stage + PD, (sb + PD) % NBUF);

    const lds_u8* tile = (const lds_u8*)smem + sb * C::BUF_BYTES;
    const lds_f4* sc = (const lds_f4*)(tile + C::STAGE_BYTES + wave * 256);

#pragma unroll
    for (int it = 0; it < C::TI / 32; ++it) {
      const int trow = it * 32;
      const int64_t tile_row0 = s_begin + (int64_t)stage * C::TI + trow;  // global streamed row of tile row 0
      if (tile_row0 >= s_end) continue;
      const bool partial = tile_row0 + 32 > s_end;                         // block-uniform
      // 16 per-register scalars of the streamed rows this lane's accumulator registers correspond to
      f32x16 sv;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 t4 = sc[(trow + 8 * q + 4 * h) >> 2];
        sv[4 * q + 0] = t4[0];
        sv[4 * q + 1] = t4[1];
        sv[4 * q + 2] = t4[2];
        sv[4 * q + 3] = t4[3];
      }
      f32x16 acc[SPW];
#pragma unroll
      for (int g = 0; g < SPW; ++g) {
        if constexpr (MODE == QM_BWD_DE) {
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[g][i] = rs[g];
        } else {
          acc[g] = sv;
        }
      }
      {
        bf16x8 af[C::KS];
#pragma unroll
        for (int s = 0; s < C::KS; ++s)
          af[s] = *(const lds_bf16x8*)(tile + (trow + r) * C::ROWB + swz2<D>(trow + r, 2 * s + h) * 16);
#pragma unroll
        for (int s = 0; s < C::KS; ++s) {
#pragma unroll
          for (int g = 0; g < SPW; ++g)
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], rf[g][s], acc[g], 0, 0, 0);
        }
        // program order: every LDS read of the tile first, then the MFMA chain (hipcc otherwise re-sinks each read
        // next to its MFMA and waits lgkmcnt(0) per k-step, exposing the LDS latency eight times per chain)
        __builtin_amdgcn_sched_group_barrier(0x100, C::KS + 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, C::KS * SPW, 0);
      }
      if (partial) {   // rows past the end of the slice: -inf scores (forward) / zero probabilities (backward)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const bool ok = tile_row0 + mfma_row(i, h) < s_end;
          if constexpr (MODE == QM_BWD_DE) {
            sv[i] = ok ? sv[i] : NEG_INF;
          } else {
#pragma unroll
            for (int g = 0; g < SPW; ++g) acc[g][i] = ok ? acc[g][i] : NEG_INF;
          }
        }
      }

      // ---------------- epilogues --------------------------------------------------------------------------
      if constexpr (MODE == QM_LSE) {
#pragma unroll
        for (int g = 0; g < SPW; ++g) {
          float tmax = acc[g][0];
#pragma unroll
          for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, acc[g][i]);
          const float mn = fmaxf(st_a[g], tmax);
          const float ms = (mn == NEG_INF) ? 0.f : mn;
          const float off = -ms * CQL_LOG2E;
          float sum = 0.f;
#pragma unroll
          for (int i = 0; i < 16; ++i) sum += fast_exp2(fmaf(acc[g][i], CQL_LOG2E, off));
          st_b[g] = st_b[g] * fast_exp2(fmaf(st_a[g], CQL_LOG2E, off)) + sum;
          st_a[g] = mn;
        }
      } else if constexpr (MODE == QM_ARGMAX) {
        // branch-free: remember the FIRST 32-row tile in which this lane's running maximum was reached; the position
        // inside that tile is resolved afterwards by qhead_argmax_resolve_kernel (one 8-MFMA chain per row)
#pragma unroll
        for (int g = 0; g < SPW; ++g) {
          float tmax = acc[g][0];
#pragma unroll
          for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, acc[g][i]);
          const bool upd = tmax > st_a[g];
          st_a[g] = upd ? tmax : st_a[g];
          st_i[g] = upd ? (int)tile_row0 : st_i[g];
        }
      } else if constexpr (TOPK) {
#pragma unroll
        for (int g = 0; g < SPW; ++g) {
          float tmax = acc[g][0];
#pragma unroll
          for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, acc[g][i]);
          // ">=": an item that ties with the k-th best but has a smaller id still belongs in front of it
          if (__builtin_amdgcn_ballot_w64(tmax >= tk_thr[g]) != 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const bool c = acc[g][i] >= tk_thr[g];
              if (__builtin_amdgcn_ballot_w64(c) != 0) {
                if (__builtin_amdgcn_ballot_w64(c && tk_cnt[g] >= QS_TOPK_BUF) != 0) tk_merge(g);
                const bool c2 = acc[g][i] >= tk_thr[g];      // the merge may have raised the bound
                if (c2) {
                  const uint32_t row = (uint32_t)(tile_row0 + mfma_row(i, h));
                  tk_buf[(g * QS_TOPK_BUF + tk_cnt[g]) * 64 + lane] =
                      ((unsigned long long)f32_order_key(acc[g][i]) << 32) | (unsigned long long)(~row);
                  tk_cnt[g] += 1;
                }
              }
            }
          }
        }
      } else if constexpr (MODE == QM_TILEMAX) {
        const int64_t tile_idx = tile_row0 >> 5;
        const bool flush = ((tile_idx + 1) % a.tg == 0) || (tile_row0 + 32 >= s_end);
#pragma unroll
        for (int g = 0; g < SPW; ++g) {
          float tmax = acc[g][0];
#pragma unroll
          for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, acc[g][i]);
          st_a[g] = fmaxf(st_a[g], tmax);
          if (flush) {
            const float v = fmaxf(st_a[g], __shfl_xor(st_a[g], 32));
            const int64_t row = res0 + g * 32 + r;
            if (h == 0 && row < a.n_res) a.tilemax[(tile_idx / a.tg) * a.n_res + row] = v;
            st_a[g] = NEG_INF;
          }
        }
      } else {  // backward modes
        // all transposed A-fragment reads of the tile are issued first: their LDS latency hides under the exp block
        bf16x8 tf[FT][2];
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) {
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              const int row = trow + 16 * s2 + 8 * jj + 4 * h + tr_q;
              const int col = ft * 32 + tr_fsub + 4 * tr_p;
              const lds_u8* p8 = tile + row * C::ROWB + swz2<D>(row, col >> 3) * 16 + (tr_p & 1) * 8;
              const bf16x4 t4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p8);
              tf[ft][s2][4 * jj + 0] = t4[0];
              tf[ft][s2][4 * jj + 1] = t4[1];
              tf[ft][s2][4 * jj + 2] = t4[2];
              tf[ft][s2][4 * jj + 3] = t4[3];
            }
          }
        }
        bf16x8 pf[SPW][2];
#pragma unroll
        for (int g = 0; g < SPW; ++g) {
          if constexpr (FUSED) {
            // Running reference m (st_a, natural units; rs = -m*log2e), shared by the two lanes of a state.  It only
            // ever moves when a tile beats it, and then jumps QS_REF_MARGIN above that tile's maximum, so the rescale
            // of the accumulators (wave-uniform branch) is rare after the first few tiles.  P = exp(S - m) <= 1.
            float tmax = acc[g][0];
#pragma unroll
            for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, acc[g][i]);
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
            tmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            const bool need = tmax > st_a[g];
            if (__builtin_amdgcn_ballot_w64(need) != 0) {
              const float nm = need ? tmax + QS_REF_MARGIN : st_a[g];
              const float f = (st_a[g] == NEG_INF) ? 0.f : fast_exp2((st_a[g] - nm) * CQL_LOG2E);   // 1 when unchanged
              st_a[g] = nm;
              rs[g] = -nm * CQL_LOG2E;
              st_b[g] *= f;
#pragma unroll
              for (int ft = 0; ft < FT; ++ft)
#pragma unroll
                for (int i = 0; i < 16; ++i) y[g][ft][i] *= f;
            }
          }
          float p[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float off = (MODE == QM_BWD_DE) ? sv[i] : rs[g];
            p[i] = fast_exp2(fmaf(acc[g][i], CQL_LOG2E, off));
          }
          if constexpr (FUSED) {
            float cs = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) cs += p[i];
            st_b[g] += cs;
          }
          if constexpr (MODE == QM_BWD_DE) {
            float cs = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) cs += p[i];
            st_b[g] += cs;
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            pf[g][0][j] = (__bf16)p[j];
            pf[g][1][j] = (__bf16)p[8 + j];
          }
        }
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) {
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int g = 0; g < SPW; ++g)
              y[g][ft] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[ft][s2], pf[g][s2], y[g][ft], 0, 0, 0);
          }
        }
        // program order: the 4*FT transposed reads, then the exp / convert block (their latency hides under it),
        // then the second MFMA chain back to back
        __builtin_amdgcn_sched_group_barrier(0x100, FT * 4, 1);
        __builtin_amdgcn_sched_group_barrier(0x402, (FUSED ? QS_FUSED_VALU : QS_BWD_VALU) * SPW, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, FT * 2 * SPW, 1);
      }
    }
   }
  }

  // ---- write partials -------------------------------------------------------------------------------------
#pragma unroll
  for (int g = 0; g < SPW; ++g) {
    const int64_t row = res0 + g * 32 + r;
    const bool ok = row < a.n_res;
    const int64_t pidx = (int64_t)split * a.n_res + row;
    if constexpr (MODE == QM_LSE) {
      const float m2 = __shfl_xor(st_a[g], 32), l2 = __shfl_xor(st_b[g], 32);
      const float M = fmaxf(st_a[g], m2);
      const float ms = (M == NEG_INF) ? 0.f : M;
      const float Lsum = st_b[g] * fast_exp2((st_a[g] - ms) * CQL_LOG2E) + l2 * fast_exp2((m2 - ms) * CQL_LOG2E);
      if (ok && h == 0) {
        a.part_a[pidx] = M;
        a.part_b[pidx] = Lsum;
      }
    } else if constexpr (TOPK) {
      tk_merge(g);
      if (ok) {
        unsigned long long* dst = a.topk_keys + (((int64_t)split * a.n_res + row) * 2 + h) * QS_TOPK_K;
#pragma unroll
        for (int j = 0; j < QS_TOPK_K; j += 2)
          *reinterpret_cast<ulonglong2*>(dst + j) = make_ulonglong2(tk_lst[g][j], tk_lst[g][j + 1]);
      }
    } else if constexpr (MODE == QM_ARGMAX) {
      const float v2 = __shfl_xor(st_a[g], 32);
      const int i2 = __shfl_xor(st_i[g], 32);
      const bool take2 = (v2 > st_a[g]) || (v2 == st_a[g] && i2 < st_i[g]);
      if (ok && h == 0) {
        a.part_a[pidx] = take2 ? v2 : st_a[g];
        a.part_i[pidx] = take2 ? i2 : st_i[g];
      }
    } else if constexpr (BWD) {
      const float sc = a.out ? a.scale : 1.0f;
      if (ok) {
        float* dst = a.out ? (a.out + row * D) : (a.slab + pidx * D);
#pragma unroll
        for (int ft = 0; ft < FT; ++ft)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float4* pd = reinterpret_cast<float4*>(dst + ft * 32 + 8 * q + 4 * h);
            float4 o = make_float4(sc * y[g][ft][4 * q + 0], sc * y[g][ft][4 * q + 1], sc * y[g][ft][4 * q + 2],
                                   sc * y[g][ft][4 * q + 3]);
            if (a.out && a.accumulate) {
              const float4 old = *pd;
              o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
            }
            *pd = o;
          }
      }
      if constexpr (MODE == QM_BWD_DE) {
        const float cs = st_b[g] + __shfl_xor(st_b[g], 32);
        if (ok && h == 0) {
          if (a.out) a.out_cs[row] = a.accumulate ? a.out_cs[row] + sc * cs : sc * cs;
          else a.slab_cs[pidx] = cs;
        }
      }
      if constexpr (FUSED) {   // (reference, sum relative to it): what qhead_finalize_lse_kernel merges
        const float ls = st_b[g] + __shfl_xor(st_b[g], 32);
        if (ok && h == 0) {
          a.part_a[pidx] = st_a[g];
          a.part_b[pidx] = ls;
        }
      }
    }
  }
}

// =============================================================================================================
// split selection + launch
// =============================================================================================================
int qs_spw_fwd(int d) {
  static const int v = qs_env_int("CQL_QS_SPW_FWD", QS_SPW_FWD);
  return (v == 4 && d <= 128) ? 4 : 2;
}

QSplit qs_choose_split(int64_t n_str, int64_t n_res, int spw, int unit_rows, int target) {
  QSplit s;
  s.rblks = (n_res + 128 * spw - 1) / (128 * spw);
  const int64_t units = (n_str + unit_rows - 1) / unit_rows;
  int64_t want = (target + s.rblks - 1) / s.rblks;
  int64_t max_split = units / 2;  // at least two units of streamed rows per slice
  if (max_split < 1) max_split = 1;
  if (want > max_split) want = max_split;
  if (want > 8) want = (want + 7) / 8 * 8;
  if (want > max_split) want = max_split;
  if (want < 1) want = 1;
  const int64_t upb = (units + want - 1) / want;
  s.split_rows = upb * unit_rows;
  s.nsplit = (int)((n_str + s.split_rows - 1) / s.split_rows);
  return s;
}

template <int D, int SPW, int MODE, int NBUF, int MINW>
static void qs_launch_n(const QArgs& a, int64_t rblks, hipStream_t s) {
  constexpr int smem = NBUF * QCfg<D>::BUF_BYTES + ((MODE == QM_TOPK || MODE == QM_TOPK10) ? 4 * SPW * QS_TOPK_BUF * 64 * 8 : 0);
  static bool attr_set_dev[CQL_MAX_DEVICES] = {};   // > 64 KiB of dynamic LDS needs the opt-in once per kernel and device
  bool& attr_set = attr_set_dev[cql_device_slot()];
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)qstream_kernel<D, SPW, MODE, NBUF, MINW>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  dim3 grid((unsigned)(rblks * a.nsplit)), block(256);
  hipLaunchKernelGGL((qstream_kernel<D, SPW, MODE, NBUF, MINW>), grid, block, smem, s, a);
}

template <int D, int SPW, int MODE>
static void qs_launch_d(const QArgs& a, int64_t rblks, hipStream_t s) {
  // ring depth: 3 stages for the single-MFMA forward modes, 2 for the two-MFMA modes (measured: -3..-6 % there, +9 %
  // on ARGMAX with 2)
  constexpr int NB = (MODE == QM_BWD_DH || MODE == QM_BWD_DE || MODE == QM_LSE_DH) ? QS_NBUF_BWD : QS_NBUF;
  qs_launch_n<D, SPW, MODE, NB, (D == 256 ? 1 : 2)>(a, rblks, s);
}

template <int MODE, int SPW>
static int qs_launch_mode(const QArgs& a, int d, int64_t rblks, hipStream_t s) {
  if constexpr (SPW == 4) {   // 128 owners per wave: d <= 128 only (register budget)
    if (d == 64) qs_launch_d<64, SPW, MODE>(a, rblks, s);
    else qs_launch_d<128, SPW, MODE>(a, rblks, s);
  } else {
    if (d == 64) qs_launch_d<64, SPW, MODE>(a, rblks, s);
    else if (d == 128) qs_launch_d<128, SPW, MODE>(a, rblks, s);
    else qs_launch_d<256, SPW, MODE>(a, rblks, s);
  }
  return 0;
}

static int qs_launch_switch(int mode, const QArgs& a, int d, int64_t rblks, hipStream_t s);
int qs_launch(int mode, const QArgs& a, int d, int64_t rblks, hipStream_t s) {
  static const int phase_of[9] = {0, CQLREC_PH_QHEAD_LSE, CQLREC_PH_QHEAD_ARGMAX, CQLREC_PH_TOPK_TILEMAX,
                                  CQLREC_PH_QHEAD_BWD_DH, CQLREC_PH_QHEAD_BWD_DE, CQLREC_PH_QHEAD_LSE,
                                  CQLREC_PH_TOPK_TILEMAX, CQLREC_PH_TOPK_TILEMAX};
  CqlProfScope prof(phase_of[mode], s);
  return qs_launch_switch(mode, a, d, rblks, s);
}
// without a measurement scope of its own (the caller brackets it)
static int qs_launch_quiet(int mode, const QArgs& a, int d, int64_t rblks, hipStream_t s) {
  return qs_launch_switch(mode, a, d, rblks, s);
}
static int qs_launch_switch(int mode, const QArgs& a, int d, int64_t rblks, hipStream_t s) {
  switch (mode) {
    case QM_LSE:
      if (qs_spw_fwd(d) == 4) return qs_launch_mode<QM_LSE, 4>(a, d, rblks, s);
      return qs_launch_mode<QM_LSE, 2>(a, d, rblks, s);
    case QM_ARGMAX:
      if (qs_spw_fwd(d) == 4) return qs_launch_mode<QM_ARGMAX, 4>(a, d, rblks, s);
      return qs_launch_mode<QM_ARGMAX, 2>(a, d, rblks, s);
    case QM_TILEMAX:
      if (qs_spw_fwd(d) == 4) return qs_launch_mode<QM_TILEMAX, 4>(a, d, rblks, s);
      return qs_launch_mode<QM_TILEMAX, 2>(a, d, rblks, s);
    case QM_TOPK: return qs_launch_mode<QM_TOPK, 2>(a, d, rblks, s);
    case QM_TOPK10: return qs_launch_mode<QM_TOPK10, 2>(a, d, rblks, s);
    case QM_BWD_DH: return qs_launch_mode<QM_BWD_DH, QS_SPW_BWD>(a, d, rblks, s);
    case QM_BWD_DE: return qs_launch_mode<QM_BWD_DE, QS_SPW_BWD>(a, d, rblks, s);
    case QM_LSE_DH: return qs_launch_mode<QM_LSE_DH, QS_SPW_BWD>(a, d, rblks, s);
  }
  return -1;
}

// =============================================================================================================
// finalize / reduce kernels
// =============================================================================================================
__global__ void qhead_finalize_lse_kernel(const float* __restrict__ pm, const float* __restrict__ pl, int nsplit,
                                          int64_t rows, float* __restrict__ lse, float* __restrict__ nlse2,
                                          float* __restrict__ nlse_nat = nullptr) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float M = NEG_INF;
#pragma unroll 8
  for (int s = 0; s < nsplit; ++s) M = fmaxf(M, pm[(int64_t)s * rows + r]);
  const float ms = (M == NEG_INF) ? 0.f : M;
  float L = 0.f;
#pragma unroll 8
  for (int s = 0; s < nsplit; ++s) L += pl[(int64_t)s * rows + r] * fast_exp2((pm[(int64_t)s * rows + r] - ms) * CQL_LOG2E);
  const float v = ms + logf(L);
  lse[r] = v;
  if (nlse2) {
    const float n2 = -v * CQL_LOG2E;
    nlse2[r] = n2;
    if (nlse_nat) nlse_nat[r] = n2 * CQL_LN2;      // exactly what qde_nlse_natural_kernel would form from nlse2
  }
}

// ARGMAX finalize: pick, per row, the slice partial with the largest maximum (ties: the earliest tile), then resolve
// the position inside that 32-item tile by recomputing its scores with the SAME MFMA chain as the streaming kernel
// (bit-identical values), and take the first item whose score equals the maximum (ties -> smallest j).
template <int D>
__global__ __launch_bounds__(64) void qhead_argmax_resolve_kernel(const float* __restrict__ pv,
                                                                  const int32_t* __restrict__ pt, int nsplit,
                                                                  int64_t rows, const uint16_t* __restrict__ H_b,
                                                                  const uint16_t* __restrict__ E_b,
                                                                  const float* __restrict__ b, int64_t n_items,
                                                                  float* __restrict__ vmax, int32_t* __restrict__ imax) {
  constexpr int KS = D / 16;
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int64_t row = blockIdx.x;
  float bv = NEG_INF;
  int bt = 0x7FFFFFFF;
  for (int s = lane; s < nsplit; s += 64) {
    const float v = pv[(int64_t)s * rows + row];
    const int t = pt[(int64_t)s * rows + row];
    if (v > bv || (v == bv && t < bt)) {
      bv = v;
      bt = t;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float v = __shfl_xor(bv, off);
    const int t = __shfl_xor(bt, off);
    if (v > bv || (v == bv && t < bt)) {
      bv = v;
      bt = t;
    }
  }
  if (lane == 0) vmax[row] = bv;
  if (!imax) return;
  if (bt == 0x7FFFFFFF) {   // nothing finite in the row
    if (lane == 0) imax[row] = 0;
    return;
  }
  const int64_t item0 = bt;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int64_t c = item0 + mfma_row(i, h);
    acc[i] = (c < n_items) ? b[c] : NEG_INF;
  }
  int64_t arow = item0 + r;
  if (arow >= n_items) arow = n_items - 1;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const bf16x8 af = *reinterpret_cast<const bf16x8*>(E_b + arow * D + 16 * s + 8 * h);
    const bf16x8 hf = *reinterpret_cast<const bf16x8*>(H_b + row * D + 16 * s + 8 * h);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, hf, acc, 0, 0, 0);
  }
  int best = 64;   // all 32 columns hold the same state: any lane pair (r, r+32) sees the 32 scores
#pragma unroll
  for (int i = 15; i >= 0; --i) best = (acc[i] == bv) ? mfma_row(i, h) : best;
  const int other = __shfl_xor(best, 32);
  best = other < best ? other : best;
  if (lane == 0) imax[row] = (int32_t)(item0 + (best < 32 ? best : 0));
}

// dst[row][f] = scale * sum_split slab[split][row][f]  (+ coef[row] * E_b[act[row]][f] when coef != NULL)
// cs_dst[row]  = scale * sum_split slab_cs[split][row]                           (when slab_cs != NULL)
template <int D>
__global__ __launch_bounds__(256) void qhead_bwd_reduce_kernel(const float* __restrict__ slab,
                                                               const float* __restrict__ slab_cs, int nsplit,
                                                               int64_t rows, float scale,
                                                               const float* __restrict__ coef,
                                                               const int32_t* __restrict__ act,
                                                               const uint16_t* __restrict__ E_b,
                                                               float* __restrict__ dst, float* __restrict__ cs_dst,
                                                               int accumulate) {
  constexpr int V = D / 4;  // float4 per row
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * V) return;
  const int64_t row = idx / V;
  const int c = (int)(idx % V);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < nsplit; ++k) {
    const float4 t = *reinterpret_cast<const float4*>(slab + ((int64_t)k * rows + row) * D + c * 4);
    s.x += t.x;
    s.y += t.y;
    s.z += t.z;
    s.w += t.w;
  }
  s.x *= scale;
  s.y *= scale;
  s.z *= scale;
  s.w *= scale;
  if (coef) {
    const float cf = coef[row];
    const uint2 e = *reinterpret_cast<const uint2*>(E_b + (int64_t)act[row] * D + c * 4);
    s.x = fmaf(cf, __uint_as_float(e.x << 16), s.x);
    s.y = fmaf(cf, __uint_as_float(e.x & 0xFFFF0000u), s.y);
    s.z = fmaf(cf, __uint_as_float(e.y << 16), s.z);
    s.w = fmaf(cf, __uint_as_float(e.y & 0xFFFF0000u), s.w);
  }
  if (accumulate) {
    const float4 old = *reinterpret_cast<const float4*>(dst + row * D + c * 4);
    s.x += old.x; s.y += old.y; s.z += old.z; s.w += old.w;
  }
  *reinterpret_cast<float4*>(dst + row * D + c * 4) = s;
  if (slab_cs && c == 0) {
    float t = 0.f;
    for (int k = 0; k < nsplit; ++k) t += slab_cs[(int64_t)k * rows + row];
    cs_dst[row] = accumulate ? cs_dst[row] + t * scale : t * scale;
  }
}

// dH from the fused forward's slabs:  dst[row][f] = scale * sum_k slab[k][row][f] * exp(m[k][row] - lse[row])
//                                                  + coef[row] * E_b[act[row]][f]
// PART 0: the whole of dH.  PART 1: the soft part only (needs the forward's lse, nothing from the loss).  PART 2: the
// one-hot term added to what PART 1 left (dst = fmaf(coef, E[a], dst)): 1 then 2 give the bits of 0.
template <int D, int PART>
__global__ __launch_bounds__(256) void qhead_dh_finish_kernel(const float* __restrict__ slab,
                                                              const float* __restrict__ pm, int nsplit, int64_t rows,
                                                              const float* __restrict__ lse, float scale,
                                                              const float* __restrict__ coef,
                                                              const int32_t* __restrict__ act,
                                                              const uint16_t* __restrict__ E_b,
                                                              float* __restrict__ dst) {
  constexpr int V = D / 4;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * V) return;
  const int64_t row = idx / V;
  const int c = (int)(idx % V);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (PART == 2) s = *reinterpret_cast<const float4*>(dst + row * D + c * 4);
  const float nl2 = (PART == 2) ? 0.f : -lse[row] * CQL_LOG2E;
  for (int k = 0; k < (PART == 2 ? 0 : nsplit); ++k) {
    const float mk = pm[(int64_t)k * rows + row];
    const float w = (mk == NEG_INF) ? 0.f : fast_exp2(fmaf(mk, CQL_LOG2E, nl2));
    const float4 t = *reinterpret_cast<const float4*>(slab + ((int64_t)k * rows + row) * D + c * 4);
    s.x = fmaf(w, t.x, s.x);
    s.y = fmaf(w, t.y, s.y);
    s.z = fmaf(w, t.z, s.z);
    s.w = fmaf(w, t.w, s.w);
  }
  if constexpr (PART != 2) {
    s.x *= scale;
    s.y *= scale;
    s.z *= scale;
    s.w *= scale;
  }
  if constexpr (PART != 1) {
    const float cf = coef[row];
    const uint2 e = *reinterpret_cast<const uint2*>(E_b + (int64_t)act[row] * D + c * 4);
    s.x = fmaf(cf, __uint_as_float(e.x << 16), s.x);
    s.y = fmaf(cf, __uint_as_float(e.x & 0xFFFF0000u), s.y);
    s.z = fmaf(cf, __uint_as_float(e.y << 16), s.z);
    s.w = fmaf(cf, __uint_as_float(e.y & 0xFFFF0000u), s.w);
  }
  *reinterpret_cast<float4*>(dst + row * D + c * 4) = s;
}

// sparse one-hot part of dQ:  g_E_out[act[b]] += coef[b] H_b[b];  g_b_out[act[b]] += coef[b]
template <int D>
__global__ __launch_bounds__(256) void qhead_bwd_sparse_kernel(const float* __restrict__ coef,
                                                               const int32_t* __restrict__ act,
                                                               const uint16_t* __restrict__ H_b, int64_t batch,
                                                               float* __restrict__ gE, float* __restrict__ gb) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= batch) return;
  const float cf = coef[b];
  const int64_t j = act[b];
#pragma unroll
  for (int k = 0; k < D / 64; ++k)
    atomicAdd(gE + j * D + k * 64 + lane, cf * bf16_bits_to_f32(H_b[b * D + k * 64 + lane]));
  if (lane == 0) atomicAdd(gb + j, cf);
}

// =============================================================================================================
// C ABI
// =============================================================================================================
static inline int64_t align256(int64_t x) { return (x + 255) / 256 * 256; }

extern "C" int64_t cqlrec_qhead_ws_bytes(int64_t rows, int64_t n_items, int32_t d) {
  (void)d;
  const QSplit sp = qs_choose_split(n_items, rows, qs_spw_fwd(d), QS_TI, QS_TARGET_BLOCKS);
  return 3 * align256((int64_t)sp.nsplit * rows * 4) + 256;
}

static int qhead_fwd_impl(const uint16_t* H_b, int64_t rows, const uint16_t* E_out_b, const float* b_out, int64_t n_items,
                         int32_t d, int32_t mode, void* ws, int64_t ws_bytes, float* out_val, int32_t* out_idx,
                         float* out_nlse2, cqlrec_stream stream, int form);
extern "C" int cqlrec_qhead_fwd(const uint16_t* H_b, int64_t rows, const uint16_t* E_out_b, const float* b_out,
                                int64_t n_items, int32_t d, int32_t mode, void* ws, int64_t ws_bytes, float* out_val,
                                int32_t* out_idx, float* out_nlse2, cqlrec_stream stream) {
  return qhead_fwd_impl(H_b, rows, E_out_b, b_out, n_items, d, mode, ws, ws_bytes, out_val, out_idx, out_nlse2, stream, 0);
}
// ARGMAX with 32 states per wave and at most 128 registers (d = 128): slower on its own than the 64-state form, but its
// waves fit beside qfwd2_kernel's on a SIMD (that kernel leaves 136 of the 512 registers per lane and 127 KiB of LDS), so
// the step driver launches the two passes together and this one's MFMAs run in the issue gaps of the other.
int cql_qhead_argmax_beside(const uint16_t* H_b, int64_t rows, const uint16_t* E_out_b, const float* b_out, int64_t n_items,
                            int32_t d, void* ws, int64_t ws_bytes, float* out_val, int32_t* out_idx, hipStream_t stream) {
  return qhead_fwd_impl(H_b, rows, E_out_b, b_out, n_items, d, CQLREC_QHEAD_ARGMAX, ws, ws_bytes, out_val, out_idx, nullptr,
                        (cqlrec_stream)stream, d == 128 ? 1 : 0);
}
// ARGMAX as the training step launches it, on the branch stream WHILE the fused forward of the other branch runs.  d = 128:
// the two-waves-per-SIMD form of the skeleton -- one of its waves (<= 128 registers) fits beside a wave of qfwd2_kernel (368)
// on a SIMD and its products run in that kernel's issue gaps, which is worth more to the step than the faster kernel:
// qargmax2_kernel alone takes 0.077 instead of 0.087 ms, the step with it 0.702 instead of 0.689 ms (same box, 2 x A/B).
// d = 256: nothing fits beside qfwd3_kernel anyway; qargmax2_kernel (1.59 instead of 2.02 ms; cfg5shard step -10 %).
int cql_qhead_argmax_step(const uint16_t* H_b, int64_t rows, const uint16_t* E_out_b, const float* b_out, int64_t n_items,
                          int32_t d, void* ws, int64_t ws_bytes, float* out_val, int32_t* out_idx, hipStream_t stream) {
  static const int force2 = getenv("CQL_QARGMAX2") && getenv("CQL_QARGMAX2")[0] == '2';      // A/B: the new kernel in the step too
  return qhead_fwd_impl(H_b, rows, E_out_b, b_out, n_items, d, CQLREC_QHEAD_ARGMAX, ws, ws_bytes, out_val, out_idx, nullptr,
                        (cqlrec_stream)stream, (d == 128 && !force2) ? 2 : 0);
}
static int qhead_fwd_impl(const uint16_t* H_b, int64_t rows, const uint16_t* E_out_b, const float* b_out, int64_t n_items,
                         int32_t d, int32_t mode, void* ws, int64_t ws_bytes, float* out_val, int32_t* out_idx,
                         float* out_nlse2, cqlrec_stream stream, int form) {
  const bool small_waves = form == 1;          // 0: the fastest form on its own; 1: small waves; 2: the skeleton's 64-state form
  CQL_REQUIRE(H_b && E_out_b && b_out && ws && out_val, "qhead_fwd: NULL pointer");
  CQL_REQUIRE(d == 64 || d == 128 || d == 256, "qhead_fwd: d=%d unsupported", d);
  CQL_REQUIRE(rows > 0 && n_items > 0, "qhead_fwd: rows=%lld n_items=%lld", (long long)rows, (long long)n_items);
  CQL_REQUIRE(mode == CQLREC_QHEAD_LSE || mode == CQLREC_QHEAD_ARGMAX, "qhead_fwd: bad mode %d", mode);
  CQL_REQUIRE(ws_bytes >= cqlrec_qhead_ws_bytes(rows, n_items, d), "qhead_fwd: workspace too small");
  const QSplit sp = small_waves ? qs_choose_split(n_items, rows, 1, QS_TI, 256)
                                : qs_choose_split(n_items, rows, qs_spw_fwd(d), QS_TI, QS_TARGET_BLOCKS);
  const int64_t seg = align256((int64_t)sp.nsplit * rows * 4);
  QArgs a = {};
  a.res = H_b;
  a.n_res = rows;
  a.str = E_out_b;
  a.n_str = n_items;
  a.str_scalar = b_out;
  a.res_scalar = nullptr;
  a.nsplit = sp.nsplit;
  a.split_rows = sp.split_rows;
  a.part_a = (float*)ws;
  a.part_b = (float*)((char*)ws + seg);
  a.part_i = (int32_t*)((char*)ws + 2 * seg);
  a.tg = 1;
  hipStream_t s = (hipStream_t)stream;
  const int thr = 256;
  if (mode == CQLREC_QHEAD_LSE) {
    qs_launch(QM_LSE, a, d, sp.rblks, s);
    CqlProfScope prof(CQLREC_PH_QHEAD_SMALL, s);
    hipLaunchKernelGGL(qhead_finalize_lse_kernel, dim3(cql_ceil_div(rows, thr)), dim3(thr), 0, s, a.part_a, a.part_b,
                       a.nsplit, rows, out_val, out_nlse2);
  } else {
    if (small_waves) {
      CqlProfScope prof(CQLREC_PH_QHEAD_ARGMAX, s);
      qs_launch_n<128, 1, QM_ARGMAX, QS_NBUF, 4>(a, sp.rblks, s);
    } else if (form == 0 && cql_qargmax2_supported(d, n_items)) {
      // one wave per SIMD, software-pipelined (qhead_argmax2.hip); fewer, longer slices than the generic form: the partials
      // fit the workspace carved above (same [slice][row] layout, fewer slices)
      int ns2;
      int64_t sr2;
      cql_qargmax2_split(rows, n_items, d, &ns2, &sr2);
      CQL_REQUIRE(ns2 <= sp.nsplit || (int64_t)ns2 * rows * 4 <= seg, "qhead_fwd: argmax partials do not fit");
      a.nsplit = ns2;
      a.split_rows = sr2;
      CqlProfScope prof(CQLREC_PH_QHEAD_ARGMAX, s);
      const int rc = cql_qargmax2_run(H_b, rows, E_out_b, b_out, n_items, d, ns2, sr2, a.part_a, a.part_i, s);
      if (rc != CQLREC_OK) return rc;
    } else {
      qs_launch(QM_ARGMAX, a, d, sp.rblks, s);
    }
    CqlProfScope prof(CQLREC_PH_QHEAD_SMALL, s);
    dim3 rg((unsigned)rows), rb(64);
#define RES_AM(DD)                                                                                                \
  hipLaunchKernelGGL(qhead_argmax_resolve_kernel<DD>, rg, rb, 0, s, a.part_a, a.part_i, a.nsplit, rows, H_b, E_out_b, \
                     b_out, n_items, out_val, out_idx)
    if (d == 64) RES_AM(64); else if (d == 128) RES_AM(128); else RES_AM(256);
#undef RES_AM
  }
  CQL_LAUNCH_CHECK("qhead_fwd");
  return CQLREC_OK;
}

extern "C" int64_t cqlrec_qhead_bwd_ws_bytes(int64_t batch, int64_t n_items, int32_t d) {
  const QSplit s1 = qs_choose_split(n_items, batch, QS_SPW_BWD, QS_TI, QS_TARGET_BLOCKS_BWD);
  const QSplit s2 = qs_choose_split(batch, n_items, QS_SPW_BWD, QS_TI, QS_TARGET_BLOCKS_BWD);
  const int64_t a1 = align256((int64_t)s1.nsplit * batch * d * 4);
  const int64_t a2 = align256((int64_t)s2.nsplit * n_items * d * 4) + align256((int64_t)s2.nsplit * n_items * 4);
  const int64_t a3 = cql_qde_ws_bytes(batch, n_items, d);
  const int64_t m = (a1 > a2 ? a1 : a2);
  return (m > a3 ? m : a3) + 256;
}

// dH only (owner = states, streamed = items)
extern "C" int cqlrec_qhead_bwd_states(const uint16_t* H_b, const float* nlse2, const float* coef, const int32_t* act,
                                       int64_t batch, const uint16_t* E_out_b, const float* b_out, int64_t n_items,
                                       int32_t d, float scale, void* ws, int64_t ws_bytes, float* dH,
                                       cqlrec_stream stream) {
  CQL_REQUIRE(H_b && nlse2 && coef && act && E_out_b && b_out && ws && dH, "qhead_bwd_states: NULL pointer");
  CQL_REQUIRE(d == 64 || d == 128 || d == 256, "qhead_bwd_states: d=%d unsupported", d);
  CQL_REQUIRE(batch > 0 && n_items > 0, "qhead_bwd_states: batch=%lld n_items=%lld", (long long)batch, (long long)n_items);
  CQL_REQUIRE(ws_bytes >= cqlrec_qhead_bwd_ws_bytes(batch, n_items, d), "qhead_bwd_states: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const QSplit sp = qs_choose_split(n_items, batch, QS_SPW_BWD, QS_TI, QS_TARGET_BLOCKS_BWD);
  QArgs a = {};
  a.res = H_b;
  a.n_res = batch;
  a.str = E_out_b;
  a.n_str = n_items;
  a.str_scalar = b_out;
  a.res_scalar = nlse2;
  a.nsplit = sp.nsplit;
  a.split_rows = sp.split_rows;
  a.slab = (float*)ws;
  a.tg = 1;
  qs_launch(QM_BWD_DH, a, d, sp.rblks, s);
  CqlProfScope prof(CQLREC_PH_QHEAD_SMALL, s);
  const int64_t n4 = batch * (d / 4);
  dim3 grid(cql_ceil_div(n4, 256)), block(256);
#define RED_DH(DD)                                                                                                  \
  hipLaunchKernelGGL(qhead_bwd_reduce_kernel<DD>, grid, block, 0, s, a.slab, (const float*)nullptr, a.nsplit, batch, \
                     scale, coef, act, E_out_b, dH, (float*)nullptr, 0)
  if (d == 64) RED_DH(64); else if (d == 128) RED_DH(128); else RED_DH(256);
#undef RED_DH
  CQL_LAUNCH_CHECK("qhead_bwd_states");
  return CQLREC_OK;
}

// g_E_out / g_b_out only (owner = items, streamed = states)
// sparse_first: scatter the one-hot part first and let the streaming kernel accumulate (callers with zeroed gradients);
// [item_lo, item_hi): item rows handled by this call (the scatter, when requested, always covers every item).
static int qhead_bwd_items_impl(const uint16_t* H_b, const float* nlse2, const float* coef, const int32_t* act,
                               int64_t batch, const uint16_t* E_out_b, const float* b_out, int64_t n_items, int32_t d,
                               float scale, void* ws, int64_t ws_bytes, float* g_E_out, float* g_b_out,
                               cqlrec_stream stream, bool sparse_first, bool do_sparse = true, int64_t item_lo = 0,
                               int64_t item_hi = -1, CqlAdamFix* defer = nullptr, const float* nlse_nat = nullptr) {
  if (defer) defer->valid = 0;
  if (item_hi < 0) item_hi = n_items;
  CQL_REQUIRE(item_lo >= 0 && item_lo < item_hi && item_hi <= n_items, "qhead_bwd_items: bad item range");
  CQL_REQUIRE(H_b && nlse2 && coef && act && E_out_b && b_out && ws && g_E_out && g_b_out, "qhead_bwd_items: NULL pointer");
  CQL_REQUIRE(d == 64 || d == 128 || d == 256, "qhead_bwd_items: d=%d unsupported", d);
  CQL_REQUIRE(batch > 0 && n_items > 0, "qhead_bwd_items: batch=%lld n_items=%lld", (long long)batch, (long long)n_items);
  CQL_REQUIRE(ws_bytes >= cqlrec_qhead_bwd_ws_bytes(batch, n_items, d), "qhead_bwd_items: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  dim3 block(256);
  auto launch_sparse = [&]() {
    CqlProfScope prof(CQLREC_PH_QHEAD_SMALL, s);
    dim3 g2(cql_ceil_div(batch, 4));
#define SP_DE(DD) hipLaunchKernelGGL(qhead_bwd_sparse_kernel<DD>, g2, block, 0, s, coef, act, H_b, batch, g_E_out, g_b_out)
    if (d == 64) SP_DE(64); else if (d == 128) SP_DE(128); else SP_DE(256);
#undef SP_DE
  };
  if (sparse_first && do_sparse) launch_sparse();
  // Large catalogues: one block per 128 items streams every state and writes its rows directly (no cross-block
  // sum).  Small catalogues: the state axis is split too, slabs are summed by the small reduce kernel.  (A third
  // variant -- a whole number of resident "rounds" first, the remainder with split states -- measured no faster:
  // the step driver instead runs this kernel concurrently with the state-side kernel, which fills the idle CUs of
  // the last round.)
  auto launch_range = [&](int64_t row0, int64_t count, bool allow_direct) {
    const QSplit sp = (allow_direct)
                          ? qs_choose_split(batch, count, QS_SPW_BWD, QS_TI, 1)
                          : qs_choose_split(batch, count, QS_SPW_BWD, QS_TI, QS_TARGET_BLOCKS_BWD);
    QArgs a = {};
    a.res = E_out_b + row0 * d;
    a.n_res = count;
    a.str = H_b;
    a.n_str = batch;
    a.str_scalar = nlse2;
    a.res_scalar = b_out + row0;
    a.nsplit = sp.nsplit;
    a.split_rows = sp.split_rows;
    a.slab = (float*)ws;
    a.slab_cs = (float*)((char*)ws + align256((int64_t)sp.nsplit * count * d * 4));
    a.tg = 1;
    const bool direct = (sp.nsplit == 1);
    if (direct) {   // one slice: the kernel scales and writes g_E_out / g_b_out itself
      a.out = g_E_out + row0 * d;
      a.out_cs = g_b_out + row0;
      a.scale = scale;
      a.accumulate = sparse_first ? 1 : 0;
    }
    qs_launch(QM_BWD_DE, a, d, sp.rblks, s);
    if (!direct) {
      CqlProfScope prof(CQLREC_PH_QHEAD_SMALL, s);
      const int64_t n4 = count * (d / 4);
      dim3 grid(cql_ceil_div(n4, 256));
#define RED_DE(DD)                                                                                            \
  hipLaunchKernelGGL(qhead_bwd_reduce_kernel<DD>, grid, block, 0, s, a.slab, a.slab_cs, a.nsplit, count, scale, \
                     (const float*)nullptr, (const int32_t*)nullptr, (const uint16_t*)nullptr,                 \
                     g_E_out + row0 * d, g_b_out + row0, sparse_first ? 1 : 0)
      if (d == 64) RED_DE(64); else if (d == 128) RED_DE(128); else RED_DE(256);
#undef RED_DE
    }
  };
  // default: the persistent, statically balanced kernel of qhead_de.hip; CQL_QDE=0 selects the generic skeleton (A/B)
  static const bool use_qde = !(getenv("CQL_QDE") && getenv("CQL_QDE")[0] == '0');
  if (use_qde) {
    const int rc = cql_qde_launch(H_b, nlse2, batch, E_out_b + item_lo * d, b_out + item_lo, item_hi - item_lo, d, scale,
                                  ws, ws_bytes, g_E_out + item_lo * d, g_b_out + item_lo, sparse_first ? 1 : 0, s,
                                  (item_lo == 0 && item_hi == n_items) ? defer : nullptr, nlse_nat);
    if (rc != CQLREC_OK) return rc;
  } else {
    const int64_t rblks_all = (n_items + 127) / 128;
    launch_range(item_lo, item_hi - item_lo, rblks_all >= QS_TARGET_BLOCKS_BWD);
  }
  if (!sparse_first && do_sparse) launch_sparse();
  CQL_LAUNCH_CHECK("qhead_bwd_items");
  return CQLREC_OK;
}

extern "C" int cqlrec_qhead_bwd_items(const uint16_t* H_b, const float* nlse2, const float* coef, const int32_t* act,
                                      int64_t batch, const uint16_t* E_out_b, const float* b_out, int64_t n_items,
                                      int32_t d, float scale, void* ws, int64_t ws_bytes, float* g_E_out,
                                      float* g_b_out, cqlrec_stream stream) {
  return qhead_bwd_items_impl(H_b, nlse2, coef, act, batch, E_out_b, b_out, n_items, d, scale, ws, ws_bytes, g_E_out,
                              g_b_out, stream, false);
}

// ---- fused forward (training): lse + slabs of the softmax-weighted item sum; see qhead_internal.h ---------------
struct FusedWs {
  QSplit sp;
  float *slab, *part_a, *part_b;
  int* flag;          // qfwd2_kernel: "a partial sum overflowed, redo the pass with the first form"
  int64_t bytes;
};
static FusedWs fused_ws(void* ws, int64_t rows, int64_t n_items, int32_t d) {
  FusedWs f;
  // item slices: 128-state row-blocks x slices = two blocks per CU for the generic form and for qfwd2_kernel (which halves
  // the row-blocks: one block per CU); qfwd3_kernel (d = 256, 128 states per block, one block per CU resident) gets half as
  // many, twice as long slices -- one round of blocks, half the slab traffic
  f.sp = qs_choose_split(n_items, rows, QS_SPW_BWD, QS_TI, cql_qfwd3_supported(d, n_items) ? QS_TARGET_BLOCKS_BWD / 2 : QS_TARGET_BLOCKS_BWD);
  const int64_t slab_b = align256((int64_t)f.sp.nsplit * rows * d * 4), seg = align256((int64_t)f.sp.nsplit * rows * 4);
  f.slab = (float*)ws;
  f.part_a = (float*)((char*)ws + slab_b);
  f.part_b = (float*)((char*)ws + slab_b + seg);
  f.flag = (int*)((char*)ws + slab_b + 2 * seg);
  f.bytes = slab_b + 2 * seg + 256;
  return f;
}

// the overflow flag of the one-wave-per-SIMD form, cleared ahead of time: a caller whose catalogue pass waits for an event
// (the item-side optimizer) clears it in front of that wait, off the critical path, and passes flag_cleared = 1
int cql_qhead_fwd_lse_dh_prepare(void* ws, int64_t rows, int64_t n_items, int32_t d, hipStream_t s) {
  if (!cql_qfwd2_supported(d, n_items) && !cql_qfwd3_supported(d, n_items)) return CQLREC_OK;
  const FusedWs f = fused_ws(ws, rows, n_items, d);
  if (hipMemsetAsync(f.flag, 0, 4, s) != hipSuccess) {
    cql_set_error("qhead_fwd_lse_dh: hipMemsetAsync failed");
    return CQLREC_ERR_HIP;
  }
  return CQLREC_OK;
}

int cql_qhead_fwd_lse_dh(const uint16_t* H_b, int64_t rows, const uint16_t* E_out_b, const float* b_out, int64_t n_items,
                         int32_t d, void* ws, int64_t ws_bytes, float* out_lse, float* out_nlse2, hipStream_t s,
                         float* out_nlse_nat, int flag_cleared) {
  CQL_REQUIRE(H_b && E_out_b && b_out && ws && out_lse, "qhead_fwd_lse_dh: NULL pointer");
  CQL_REQUIRE(d == 64 || d == 128 || d == 256, "qhead_fwd_lse_dh: d=%d unsupported", d);
  CQL_REQUIRE(rows > 0 && n_items > 0, "qhead_fwd_lse_dh: rows=%lld n_items=%lld", (long long)rows, (long long)n_items);
  const FusedWs f = fused_ws(ws, rows, n_items, d);
  CQL_REQUIRE(ws_bytes >= f.bytes, "qhead_fwd_lse_dh: workspace too small");
  QArgs a = {};
  a.res = H_b;
  a.n_res = rows;
  a.str = E_out_b;
  a.n_str = n_items;
  a.str_scalar = b_out;
  a.nsplit = f.sp.nsplit;
  a.split_rows = f.sp.split_rows;
  a.slab = f.slab;
  a.part_a = f.part_a;
  a.part_b = f.part_b;
  a.tg = 1;
  if (cql_qfwd2_supported(d, n_items) || cql_qfwd3_supported(d, n_items)) {
    // one wave per SIMD, fixed per-slice reference (qhead_fwd2.hip, qhead_fwd3.hip); the first form follows, guarded by the flag the
    // second sets when a partial sum overflowed -- its blocks return at once otherwise
    if (!flag_cleared && hipMemsetAsync(f.flag, 0, 4, s) != hipSuccess) {
      cql_set_error("qhead_fwd_lse_dh: hipMemsetAsync failed");
      return CQLREC_ERR_HIP;
    }
    QFwd2Args a2 = {};
    a2.H_b = H_b;
    a2.n_states = rows;
    a2.E_b = E_out_b;
    a2.bias = b_out;
    a2.n_items = n_items;
    a2.nsplit = f.sp.nsplit;
    a2.split_rows = f.sp.split_rows;
    a2.slab = f.slab;
    a2.part_a = f.part_a;
    a2.part_b = f.part_b;
    a2.flag = f.flag;
    {
      CqlProfScope prof(CQLREC_PH_QHEAD_LSE, s);
      const int rc = (d == 256) ? cql_qfwd3_run(a2, d, s) : cql_qfwd2_run(a2, d, s);
      if (rc != CQLREC_OK) return rc;
    }
    a.guard = f.flag;
    CqlProfScope prof(CQLREC_PH_QHEAD_SMALL, s);
    qs_launch_quiet(QM_LSE_DH, a, d, f.sp.rblks, s);
  } else {
    qs_launch(QM_LSE_DH, a, d, f.sp.rblks, s);
  }
  CqlProfScope prof(CQLREC_PH_QHEAD_SMALL, s);
  hipLaunchKernelGGL(qhead_finalize_lse_kernel, dim3(cql_ceil_div(rows, 256)), dim3(256), 0, s, a.part_a, a.part_b,
                     a.nsplit, rows, out_lse, out_nlse2, out_nlse_nat);
  CQL_LAUNCH_CHECK("qhead_fwd_lse_dh");
  return CQLREC_OK;
}

int cql_qhead_dh_finish(const void* ws, int64_t rows, int64_t n_items, int32_t d, const float* lse, const float* coef,
                        const int32_t* act, const uint16_t* E_out_b, float scale, float* dH, hipStream_t s, int part) {
  CQL_REQUIRE(ws && lse && act && E_out_b && dH && (coef || part == 1) && part >= 0 && part <= 2, "qhead_dh_finish: bad arguments");
  const FusedWs f = fused_ws(const_cast<void*>(ws), rows, n_items, d);
  CqlProfScope prof(CQLREC_PH_QHEAD_SMALL, s);
  const int64_t n4 = rows * (d / 4);
  dim3 grid(cql_ceil_div(n4, 256)), block(256);
#define FIN_DH(DD, PP)                                                                                                   \
  hipLaunchKernelGGL((qhead_dh_finish_kernel<DD, PP>), grid, block, 0, s, f.slab, f.part_a, f.sp.nsplit, rows, lse, scale, \
                     coef, act, E_out_b, dH)
#define FIN_DH_D(PP)                                                                                                     \
  do {                                                                                                                   \
    if (d == 64) FIN_DH(64, PP); else if (d == 128) FIN_DH(128, PP); else FIN_DH(256, PP);                               \
  } while (0)
  if (part == 0) FIN_DH_D(0); else if (part == 1) FIN_DH_D(1); else FIN_DH_D(2);
#undef FIN_DH_D
#undef FIN_DH
  CQL_LAUNCH_CHECK("qhead_dh_finish");
  return CQLREC_OK;
}

extern "C" int64_t cqlrec_qhead_fused_ws_bytes(int64_t rows, int64_t n_items, int32_t d) {
  return fused_ws(nullptr, rows, n_items, d).bytes + 256;
}
extern "C" int cqlrec_qhead_fwd_lse_dh(const uint16_t* H_b, int64_t rows, const uint16_t* E_out_b, const float* b_out,
                                       int64_t n_items, int32_t d, void* ws, int64_t ws_bytes, float* out_lse,
                                       float* out_nlse2, cqlrec_stream stream) {
  return cql_qhead_fwd_lse_dh(H_b, rows, E_out_b, b_out, n_items, d, ws, ws_bytes, out_lse, out_nlse2, (hipStream_t)stream,
                              nullptr);
}
extern "C" int cqlrec_qhead_dh_finish(const void* ws, int64_t rows, int64_t n_items, int32_t d, const float* lse,
                                      const float* coef, const int32_t* act, const uint16_t* E_out_b, float scale,
                                      float* dH, cqlrec_stream stream) {
  return cql_qhead_dh_finish(ws, rows, n_items, d, lse, coef, act, E_out_b, scale, dH, (hipStream_t)stream);
}

int cql_qhead_bwd_items_long(const uint16_t* H_b, const float* nlse2, const float* coef, const int32_t* act, int64_t batch,
                             const uint16_t* E_out_b, const float* b_out, int64_t n_items, int32_t d, float scale, void* ws,
                             int64_t ws_bytes, float* g_E_out, float* g_b_out, hipStream_t stream, CqlAdamFix* defer,
                             const float* nlse_nat) {
  return qhead_bwd_items_impl(H_b, nlse2, coef, act, batch, E_out_b, b_out, n_items, d, scale, ws, ws_bytes, g_E_out,
                              g_b_out, (cqlrec_stream)stream, false, false, 0, -1, defer, nlse_nat);
}

int cql_qhead_bwd_items_acc(const uint16_t* H_b, const float* nlse2, const float* coef, const int32_t* act, int64_t batch,
                            const uint16_t* E_out_b, const float* b_out, int64_t n_items, int32_t d, float scale, void* ws,
                            int64_t ws_bytes, float* g_E_out, float* g_b_out, hipStream_t stream, int do_sparse,
                            int64_t item_lo, int64_t item_hi, CqlAdamFix* defer, const float* nlse_nat) {
  return qhead_bwd_items_impl(H_b, nlse2, coef, act, batch, E_out_b, b_out, n_items, d, scale, ws, ws_bytes, g_E_out,
                              g_b_out, (cqlrec_stream)stream, true, do_sparse != 0, item_lo, item_hi, defer, nlse_nat);
}

extern "C" int cqlrec_qhead_bwd(const uint16_t* H_b, const float* nlse2, const float* coef, const int32_t* act,
                                int64_t batch, const uint16_t* E_out_b, const float* b_out, int64_t n_items, int32_t d,
                                float scale, void* ws, int64_t ws_bytes, float* dH, float* g_E_out, float* g_b_out,
                                cqlrec_stream stream) {
  int rc = cqlrec_qhead_bwd_items(H_b, nlse2, coef, act, batch, E_out_b, b_out, n_items, d, scale, ws, ws_bytes, g_E_out,
                                  g_b_out, stream);
  if (rc != CQLREC_OK) return rc;
  return cqlrec_qhead_bwd_states(H_b, nlse2, coef, act, batch, E_out_b, b_out, n_items, d, scale, ws, ws_bytes, dH,
                                 stream);
}
