// Shared device/host helpers for the cqlrec HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/cqlrec.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;   // 32x32 accumulator (16 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

#define CQL_WAVE 64
#define CQL_LOG2E 1.4426950408889634f
#define CQL_LN2 0.6931471805599453f
#define NEG_INF_F (-__builtin_inff())

// ---- error plumbing (host) ---------------------------------------------------------------------------------
void cql_set_error(const char* fmt, ...);
#define CQL_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      cql_set_error(__VA_ARGS__);         \
      return CQLREC_ERR_INVALID;          \
    }                                     \
  } while (0)
#define CQL_LAUNCH_CHECK(name)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      cql_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return CQLREC_ERR_HIP;                                                \
    }                                                                       \
  } while (0)

// index of the calling thread's current device into small per-device state tables (internal streams, "opt-in done"
// flags): a process that drives several devices gets one set per device instead of silently sharing the first one's
#define CQL_MAX_DEVICES 16
static inline int cql_device_slot() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  return dev % CQL_MAX_DEVICES;
}

// ---- measurement hooks (see cqlrec_prof_enable) -------------------------------------------------------------
void cql_prof_begin(int phase, hipStream_t s);
void cql_prof_end(hipStream_t s);
struct CqlProfScope {
  hipStream_t s;
  CqlProfScope(int phase, hipStream_t st) : s(st) { cql_prof_begin(phase, st); }
  ~CqlProfScope() { cql_prof_end(s); }
};

// deterministic one-hot scatter of the Q-head backward (gbwd.hip): sort of (act[b], b) ahead of time, then a segmented sum
int cql_encoder_bwd_parts(const float* dH, const uint16_t* z_b, const uint16_t* h0_b, const uint16_t* W1_b,
                          const uint16_t* W2_b, int64_t rows, int32_t d, void* ws, int64_t ws_bytes, float* g_W1, float* g_b1,
                          float* g_W2, float* g_b2, float* dh0, int parts, hipStream_t s);
// misc.hip: cqlrec_td_loss in two launches (coefficients by many blocks; the loss value by one, off the critical path)
int cql_td_coef(const float* q_a, const float* lse, const float* q_targ, const float* rew, const float* done, int32_t batch,
                float gamma, float alpha, float inv_batch, float* coef, float* y, float* term, hipStream_t s);
int cql_td_loss_sum(const float* term, int32_t batch, float inv_batch, float* loss_out, hipStream_t s);
int64_t cql_onehot_ws_bytes(int64_t batch, int32_t d);
int cql_onehot_prepare(const int32_t* act, int64_t batch, int64_t n_items, int32_t d, void* ws, int64_t ws_bytes,
                       hipStream_t s);
int cql_onehot_apply(const float* coef, const uint16_t* H_b, int64_t batch, int64_t n_items, int32_t d, void* ws,
                     float* g_E_out, float* g_b_out, hipStream_t s, int accumulate = 0);

static inline int cql_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- bf16 <-> f32 ------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE; NaN stays NaN) -- identical to the oracle's RNE on finite values
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float bf16_round_f32(float f) { return bf16_bits_to_f32(f32_to_bf16_bits(f)); }

// 8 bf16 packed in a uint4 -> 8 floats
__device__ __forceinline__ void unpack_bf16x8(const uint4& v, float (&o)[8]) {
  o[0] = __uint_as_float(v.x << 16);
  o[1] = __uint_as_float(v.x & 0xFFFF0000u);
  o[2] = __uint_as_float(v.y << 16);
  o[3] = __uint_as_float(v.y & 0xFFFF0000u);
  o[4] = __uint_as_float(v.z << 16);
  o[5] = __uint_as_float(v.z & 0xFFFF0000u);
  o[6] = __uint_as_float(v.w << 16);
  o[7] = __uint_as_float(v.w & 0xFFFF0000u);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return (uint32_t)f32_to_bf16_bits(lo) | ((uint32_t)f32_to_bf16_bits(hi) << 16);
}

// ---- MFMA 32x32x16 bf16 fragment geometry (cdna_hip_programming.md section 3) ------------------------------
//  A: lane l holds A[row l&31][k = 8*(l>>5) + j], j=0..7      B: lane l holds B[k = 8*(l>>5)+j][col l&31]
//  C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5)
__device__ __forceinline__ int mfma_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// XOR swizzle of the 16-byte chunk index inside an LDS row of D bf16 so that ds_read_b128 by 16 lanes with
// distinct rows (mod 16) and equal chunk is bank-conflict free (T2).  D=64: rows are 128 B (two per bank row).
template <int D>
__device__ __forceinline__ int swz_chunk(int row, int ch) {
  if constexpr (D == 64) return ch ^ ((row >> 1) & 7);
  else return ch ^ (row & 15);
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// order-preserving map float -> uint32 (larger float <=> larger key; -inf smallest)
__device__ __forceinline__ uint32_t f32_order_key(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_from_order_key(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __uint_as_float(u);
}

// Deferred sum of the cut pieces of the item-side backward (qhead_de.hip).  The persistent kernel writes the piece of a
// cut item group that does not hold stage 0 to a slab; qde_fixup_kernel adds those slabs to the gradient rows.  The
// single-rank step driver skips that launch: the Adam launch that consumes g_E_out / g_b_out right behind it adds the
// same slabs, in the same order, while it reads the gradient (cql_adam_ema_fix) -- one launch and one read-modify-write
// of the cut rows less on the step's critical path.
struct CqlAdamFix {
  int valid;                // 0: nothing deferred (the gradient is complete)
  const float* slab;        // [nblk][items x D]
  const float* slab_cs;     // [nblk][items]
  int32_t G, T, nblk, items, D;
  float scale;
  int64_t n_items;
  int64_t rows_off, cs_off; // element offsets of g_E_out / g_b_out inside the updated range
};
int cql_adam_ema_fix(float* theta, float* grads, float* m, float* v, float* target, uint16_t* theta_b, uint16_t* target_b,
                     int64_t n, float step_size, float sqrt_bc2, float beta1, float beta2, float eps, float tau,
                     int32_t zero_grads, const CqlAdamFix* fix, hipStream_t stream);
