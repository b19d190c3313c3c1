// Shared pieces of the item-side backward kernels (qhead_de.hip: 32 items per wave, two waves per SIMD; qhead_de2.hip:
// 64 items per wave, one wave per SIMD): LDS image geometry, LDS-DMA through buffer loads, kernel arguments.
#pragma once
#include "qhead_internal.h"

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(3))) f32x4 lds_f4;
typedef __attribute__((address_space(3))) bf16x8 lds_bf16x8;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define QDE_MAX_ITEMS 256  // items per group: 32 per wave, 4 or 8 waves per block
#ifndef QDE_VALU_HINT
#define QDE_VALU_HINT 32   // VALU instructions the scheduler is asked to place between the transposed reads and the 2nd chain
#endif

template <int D, int WAVES = 4>
struct DeCfg {
  static constexpr int ITEMS = 32 * WAVES;                   // items per group
  static constexpr int ROWB = 2 * D;
  static constexpr int KS = D / 16;
  static constexpr int FT = D / 32;
  static constexpr int TI = (D == 256) ? 32 : 64;            // states per stage
  static constexpr int TILES = TI / 32;
  static constexpr int STAGE_BYTES = TI * ROWB;              // 8 KiB (d=64) / 16 KiB
  static constexpr int LPS = STAGE_BYTES / 1024 / WAVES;     // LDS-DMA pieces (1 KiB) per wave per stage
  static constexpr int PPG = D / 64;                         // pieces per 8-row group
  static constexpr int RG_BYTES = 16 * D;                    // one 8-row group of the LDS image
  static constexpr int TILE_BYTES = 4 * RG_BYTES;            // 32 rows
  static constexpr int STRIP_BYTES = 256;                    // -lse*log2e of the stage's states: ONE copy per stage, loaded
                                                             // by wave (stage % WAVES)
  static constexpr int BUF_BYTES = STAGE_BYTES + STRIP_BYTES;
  static constexpr int PSTEP = WAVES * 1024;                 // source bytes between a wave's consecutive pieces
};

struct QDeArgs {
  const uint16_t* H_b;      // [n_states x D] streamed
  const float* nlse2;       // [n_states]  -lse * log2e
  int64_t n_states;
  const uint16_t* E_b;      // [n_items x D] owner rows (already offset to the first item of this call)
  const float* bias;        // [n_items]
  int64_t n_items;
  float* out;               // [n_items x D]
  float* out_cs;            // [n_items]
  float scale;
  int accumulate;           // out / out_cs hold the one-hot part already
  float* slab;              // [grid][ITEMS x D]   pieces that do not hold stage 0 of their group
  float* slab_cs;           // [grid][ITEMS]
  int32_t G, T;             // item groups, stages per group
  unsigned long long* stamps;   // diagnostic (CQL_QDE_STAMPS=1): per block {shader-clock ticks, 100 MHz ticks} of its run; else NULL
};

// clock stamps of a diagnostic run: written to a buffer no other code reads (never part of an output)
__device__ __forceinline__ void qde_stamp(unsigned long long& tk, unsigned long long& rt) {
  tk = __builtin_amdgcn_s_memtime();
  rt = __builtin_amdgcn_s_memrealtime();
}

__device__ __forceinline__ uint32_t lds_addr_of(const void* p) { return (uint32_t)(uintptr_t)(lds_void_t*)p; }

// LDS-DMA: 64 lanes x 16 B -> 1 KiB at LDS byte `lds_dst` (wave-uniform), source = descriptor base + voff (per lane) + soff
__device__ __forceinline__ void bdma16(uint32_t voff, __amdgpu_buffer_rsrc_t rsrc, uint32_t soff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void bdma4(uint32_t voff, __amdgpu_buffer_rsrc_t rsrc, uint32_t soff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}

template <int N>
__device__ __forceinline__ void de_wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else static_assert(N < 0, "add the vmcnt literal");
}


// qhead_de2.hip: 64 items per wave, one wave per SIMD (d = 64, 128).  `a.nlse2` holds -lse in NATURAL units there.
int cql_qde2_run(const QDeArgs& a, int d, int grid, hipStream_t s);
// qhead_de3.hip: 32 items per wave, one wave per SIMD, in-wave pipeline (d = 256).  -lse in NATURAL units as for qde2.
bool cql_qde3_supported(int d, int64_t batch);
int cql_qde3_run(const QDeArgs& a, int d, int grid, hipStream_t s);
