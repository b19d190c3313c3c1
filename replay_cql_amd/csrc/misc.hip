// Sampler, window gather (+backward), encoder layers (+backward), TD loss, gather-dot, fused Adam.
// gfx950 only.  Algorithmic bytes per unit are listed in DESIGN.md.
#include <stdlib.h>
#include "common.h"
#include "adam_math.h"

// =============================================================================================================
// error plumbing
// =============================================================================================================
static thread_local char g_err[512] = "";
void cql_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* cqlrec_last_error(void) { return g_err; }
extern "C" int cqlrec_abi_version(void) { return CQLREC_ABI_VERSION; }

// =============================================================================================================
// measurement hooks: event pairs around kernels, on the launching stream
// =============================================================================================================
#define CQL_PROF_POOL 8192
static struct {
  bool on = false;
  bool created = false;
  hipEvent_t ev[CQL_PROF_POOL][2];
  int phase[CQL_PROF_POOL];
  int n = 0;
  int open = -1;
  uint32_t mask = 0xFFFFFFFFu;   // phases that are bracketed while on
} g_prof;

void cql_prof_begin(int phase, hipStream_t s) {
  if (!g_prof.on || g_prof.n >= CQL_PROF_POOL || !((g_prof.mask >> phase) & 1u)) return;
  g_prof.open = g_prof.n++;
  g_prof.phase[g_prof.open] = phase;
  (void)hipEventRecord(g_prof.ev[g_prof.open][0], s);
}
void cql_prof_end(hipStream_t s) {
  if (!g_prof.on || g_prof.open < 0) return;
  (void)hipEventRecord(g_prof.ev[g_prof.open][1], s);
  g_prof.open = -1;
}
extern "C" int cqlrec_prof_select(uint32_t phase_mask) {
  g_prof.mask = phase_mask;
  return CQLREC_OK;
}
extern "C" int cqlrec_prof_enable(int32_t on) {
  if (on && !g_prof.created) {
    for (int i = 0; i < CQL_PROF_POOL; ++i)
      for (int k = 0; k < 2; ++k)
        if (hipEventCreate(&g_prof.ev[i][k]) != hipSuccess) {
          cql_set_error("prof_enable: hipEventCreate failed");
          return CQLREC_ERR_HIP;
        }
    g_prof.created = true;
  }
  g_prof.on = on != 0;
  g_prof.n = 0;
  g_prof.open = -1;
  return CQLREC_OK;
}
extern "C" int cqlrec_prof_read(double* ms_sum, int64_t* launches) {
  CQL_REQUIRE(ms_sum && launches, "prof_read: NULL pointer");
  for (int p = 0; p < CQLREC_PH_COUNT; ++p) {
    ms_sum[p] = 0.0;
    launches[p] = 0;
  }
  for (int i = 0; i < g_prof.n; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(g_prof.ev[i][1]) != hipSuccess ||
        hipEventElapsedTime(&ms, g_prof.ev[i][0], g_prof.ev[i][1]) != hipSuccess) {
      cql_set_error("prof_read: event query failed");
      return CQLREC_ERR_HIP;
    }
    ms_sum[g_prof.phase[i]] += ms;
    launches[g_prof.phase[i]] += 1;
  }
  g_prof.n = 0;
  return CQLREC_OK;
}

extern "C" int cqlrec_layout_make(int64_t n_items, int32_t d, cqlrec_layout* out) {
  CQL_REQUIRE(out != nullptr, "layout_make: out is NULL");
  CQL_REQUIRE(n_items > 0 && n_items < (1ll << 31) - 64, "layout_make: n_items=%lld out of range", (long long)n_items);
  CQL_REQUIRE(d == 64 || d == 128 || d == 256, "layout_make: d=%d unsupported (64, 128 or 256)", d);
  int64_t sizes[7] = {(n_items + 1) * d, n_items * d, n_items, (int64_t)d * d, d, (int64_t)d * d, d};
  int64_t offs[7], cur = 0;
  for (int i = 0; i < 7; ++i) {
    offs[i] = cur;
    cur += (sizes[i] + CQLREC_SEG_ALIGN - 1) / CQLREC_SEG_ALIGN * CQLREC_SEG_ALIGN;
  }
  out->n_items = n_items;
  out->d = d;
  out->reserved = 0;
  out->off_E_in = offs[0];
  out->off_E_out = offs[1];
  out->off_b_out = offs[2];
  out->off_W1 = offs[3];
  out->off_b1 = offs[4];
  out->off_W2 = offs[5];
  out->off_b2 = offs[6];
  out->total = cur;
  return CQLREC_OK;
}

// =============================================================================================================
// sampler
// =============================================================================================================
__host__ __device__ static inline uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ void sample_kernel(const int64_t* __restrict__ offsets, const int32_t* __restrict__ items,
                              const float* __restrict__ rewards, int64_t n_users, uint64_t k1, uint64_t slot0,
                              int batch, int32_t* users, int32_t* tpos, int32_t* act, float* rew, float* done) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const uint64_t nnz = (uint64_t)offsets[n_users];
  const uint64_t k2 = mix64(k1 + slot0 + (uint64_t)b);
  const int64_t p = (int64_t)__umul64hi(k2, nnz);
  int64_t lo = 0, hi = n_users;  // offsets[lo] <= p < offsets[hi]
  while (hi - lo > 1) {
    int64_t mid = (lo + hi) >> 1;
    if (offsets[mid] <= p) lo = mid; else hi = mid;
  }
  const int64_t o0 = offsets[lo], o1 = offsets[lo + 1];
  users[b] = (int32_t)lo;
  tpos[b] = (int32_t)(p - o0);
  act[b] = items[p];
  rew[b] = rewards[p];
  done[b] = (p == o1 - 1) ? 1.0f : 0.0f;
}

extern "C" int cqlrec_sample_transitions(const int64_t* offsets, const int32_t* items, const float* rewards,
                                         int64_t n_users, uint64_t seed, uint64_t step, uint64_t slot0,
                                         int32_t batch, int32_t* users, int32_t* tpos, int32_t* act, float* rew,
                                         float* done, cqlrec_stream stream) {
  CQL_REQUIRE(offsets && items && rewards && users && tpos && act && rew && done, "sample_transitions: NULL pointer");
  CQL_REQUIRE(n_users > 0 && batch > 0, "sample_transitions: n_users=%lld batch=%d", (long long)n_users, batch);
  const uint64_t k1 = mix64(seed ^ (step * 0xD1B54A32D192ED03ull));
  CqlProfScope prof(CQLREC_PH_SAMPLE, (hipStream_t)stream);
  hipLaunchKernelGGL(sample_kernel, dim3(cql_ceil_div(batch, 256)), dim3(256), 0, (hipStream_t)stream, offsets, items,
                     rewards, n_users, k1, slot0, batch, users, tpos, act, rew, done);
  CQL_LAUNCH_CHECK("sample_transitions");
  return CQLREC_OK;
}

// =============================================================================================================
// window gather + masked mean.  One wave per state; a row of D bf16 is D/8 lanes x 16 B, so one wave-instruction
// fetches 64/(D/8) whole rows (1 KiB, fully coalesced per row).  Up to 4 such instructions are kept in flight.
// =============================================================================================================
template <int D>
__global__ __launch_bounds__(256) void gather_pool_fwd_kernel(const uint16_t* __restrict__ E_in_b,
                                                              const int64_t* __restrict__ offsets,
                                                              const int32_t* __restrict__ items,
                                                              const int32_t* __restrict__ users,
                                                              const int32_t* __restrict__ ends, int end_delta,
                                                              int64_t n_states, int L, float* __restrict__ h0,
                                                              uint16_t* __restrict__ h0_b, int32_t* __restrict__ lens) {
  constexpr int LPR = D / 8;         // lanes per row
  constexpr int RPI = 64 / LPR;      // rows per wave-instruction
  const int lane = threadIdx.x & 63;
  const int64_t state = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (state >= n_states) return;
  const int u = users[state];
  const int64_t o0 = offsets[u];
  const int end = ends ? (ends[state] + end_delta) : (int)(offsets[u + 1] - o0);
  const int len = end < L ? end : L;
  const int32_t* win = items + o0 + end - len;
  const int slot = lane / LPR, c = lane % LPR;

  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;

  for (int j0 = 0; j0 < len; j0 += 4 * RPI) {
    int it[4];
    uint4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = j0 + q * RPI + slot;
      it[q] = (j < len) ? win[j] : -1;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (it[q] >= 0) v[q] = *reinterpret_cast<const uint4*>(E_in_b + (int64_t)it[q] * D + c * 8);
      else v[q] = make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float f[8];
      unpack_bf16x8(v[q], f);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += f[i];
    }
  }
  // combine the RPI row slots (lanes with equal c)
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += __shfl_xor(acc[i], off);
  }
  if (slot == 0) {
    const float fl = (float)len;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (len > 0) ? acc[i] / fl : 0.f;
    if (h0) {
      float4* o = reinterpret_cast<float4*>(h0 + state * D + c * 8);
      o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
      o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
    if (h0_b) {
      uint4 p;
      p.x = pack_bf16x2(acc[0], acc[1]);
      p.y = pack_bf16x2(acc[2], acc[3]);
      p.z = pack_bf16x2(acc[4], acc[5]);
      p.w = pack_bf16x2(acc[6], acc[7]);
      *reinterpret_cast<uint4*>(h0_b + state * D + c * 8) = p;
    }
    if (lens && c == 0) lens[state] = len;
  }
}

extern "C" int cqlrec_gather_pool_fwd(const uint16_t* E_in_b, const int64_t* offsets, const int32_t* items,
                                      const int32_t* users, const int32_t* ends, int32_t end_delta, int64_t n_states,
                                      int32_t L, int32_t d, float* h0, uint16_t* h0_b, int32_t* lens,
                                      cqlrec_stream stream) {
  CQL_REQUIRE(E_in_b && offsets && items && users, "gather_pool_fwd: NULL pointer");
  CQL_REQUIRE(n_states >= 0 && L > 0, "gather_pool_fwd: n_states=%lld L=%d", (long long)n_states, L);
  CQL_REQUIRE(d == 64 || d == 128 || d == 256, "gather_pool_fwd: d=%d unsupported", d);
  if (n_states == 0) return CQLREC_OK;
  dim3 grid(cql_ceil_div(n_states, 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  CqlProfScope prof(CQLREC_PH_GATHER_FWD, s);
#define GP_LAUNCH(DD)                                                                                          \
  hipLaunchKernelGGL(gather_pool_fwd_kernel<DD>, grid, block, 0, s, E_in_b, offsets, items, users, ends, end_delta, \
                     n_states, L, h0, h0_b, lens)
  if (d == 64) GP_LAUNCH(64); else if (d == 128) GP_LAUNCH(128); else GP_LAUNCH(256);
#undef GP_LAUNCH
  CQL_LAUNCH_CHECK("gather_pool_fwd");
  return CQLREC_OK;
}

// backward: one wave per state, each wave-instruction adds 256 contiguous bytes of one row (Guideline 12 shape).
template <int D>
__global__ __launch_bounds__(256) void gather_pool_bwd_kernel(const float* __restrict__ dh0,
                                                              const int64_t* __restrict__ offsets,
                                                              const int32_t* __restrict__ items,
                                                              const int32_t* __restrict__ users,
                                                              const int32_t* __restrict__ ends, int end_delta,
                                                              int64_t n_states, int L, float* __restrict__ g_E_in) {
  constexpr int PER = D / 64;
  const int lane = threadIdx.x & 63;
  const int64_t state = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (state >= n_states) return;
  const int u = users[state];
  const int64_t o0 = offsets[u];
  const int end = ends ? (ends[state] + end_delta) : (int)(offsets[u + 1] - o0);
  const int len = end < L ? end : L;
  if (len == 0) return;
  const int32_t* win = items + o0 + end - len;
  float g[PER];
  const float fl = (float)len;
#pragma unroll
  for (int k = 0; k < PER; ++k) g[k] = dh0[state * D + k * 64 + lane] / fl;
  for (int j = 0; j < len; ++j) {
    float* row = g_E_in + (int64_t)win[j] * D;
#pragma unroll
    for (int k = 0; k < PER; ++k) atomicAdd(row + k * 64 + lane, g[k]);
  }
}

extern "C" int cqlrec_gather_pool_bwd(const float* dh0, const int64_t* offsets, const int32_t* items,
                                      const int32_t* users, const int32_t* ends, int32_t end_delta, int64_t n_states,
                                      int32_t L, int32_t d, float* g_E_in, cqlrec_stream stream) {
  CQL_REQUIRE(dh0 && offsets && items && users && g_E_in, "gather_pool_bwd: NULL pointer");
  CQL_REQUIRE(d == 64 || d == 128 || d == 256, "gather_pool_bwd: d=%d unsupported", d);
  if (n_states <= 0) return CQLREC_OK;
  dim3 grid(cql_ceil_div(n_states, 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  CqlProfScope prof(CQLREC_PH_GATHER_BWD, s);
#define GB_LAUNCH(DD)                                                                                              \
  hipLaunchKernelGGL(gather_pool_bwd_kernel<DD>, grid, block, 0, s, dh0, offsets, items, users, ends, end_delta, \
                     n_states, L, g_E_in)
  if (d == 64) GB_LAUNCH(64); else if (d == 128) GB_LAUNCH(128); else GB_LAUNCH(256);
#undef GB_LAUNCH
  CQL_LAUNCH_CHECK("gather_pool_bwd");
  return CQLREC_OK;
}

// =============================================================================================================
// encoder layer on bf16 MFMA:  Y[m][o] = act(sum_k X[m][k] W[o][k] + bias[o]).  One wave = 32 rows x all D outputs;
// X fragments stay in registers, W (d x d bf16, L2-resident) is read fragment-wise.  <0.1 % of the step's flops.
// =============================================================================================================
template <int D>
__global__ __launch_bounds__(64) void linear_bf16_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ W,
                                                         const float* __restrict__ bias, int64_t rows, int relu,
                                                         float* __restrict__ Y, uint16_t* __restrict__ Yb) {
  constexpr int KS = D / 16;
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * 32;
  int64_t mrow = m0 + r;
  if (mrow >= rows) mrow = rows - 1;
  bf16x8 xf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) xf[s] = *reinterpret_cast<const bf16x8*>(X + mrow * D + 16 * s + 8 * h);
#pragma unroll 1
  for (int ot = 0; ot < D / 32; ++ot) {
    const int o = ot * 32 + r;
    const float bo = bias[o];
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = bo;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bf16x8 wf = *reinterpret_cast<const bf16x8*>(W + (int64_t)o * D + 16 * s + 8 * h);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[s], wf, acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t m = m0 + mfma_row(i, h);
      float v = acc[i];
      if (relu) v = fmaxf(v, 0.f);
      if (m < rows) {
        if (Y) Y[m * D + o] = v;
        if (Yb) Yb[m * D + o] = f32_to_bf16_bits(v);
      }
    }
  }
}

// Both encoder layers in one launch: z = relu(X W1^T + b1) (bf16, kept: the backward reads it), h = z W2^T + b2 (bf16).  The
// products, their k-order and the roundings are those of two linear_bf16_kernel launches -- the bf16 z tile of the block's
// 32 rows goes through LDS (accumulator layout -> operand layout) instead of through memory and a second launch.
template <int D>
__global__ __launch_bounds__(64) void encoder_fwd_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ W1,
                                                         const float* __restrict__ b1, const uint16_t* __restrict__ W2,
                                                         const float* __restrict__ b2, int64_t rows,
                                                         uint16_t* __restrict__ Zb, uint16_t* __restrict__ Hb) {
  constexpr int KS = D / 16;
  constexpr int LDZ = D + 8;                       // row stride of the z tile in bf16 (16-byte aligned rows, skewed banks)
  __shared__ __attribute__((aligned(16))) uint16_t zt[32 * LDZ];
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * 32;
  int64_t mrow = m0 + r;
  if (mrow >= rows) mrow = rows - 1;
  bf16x8 xf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) xf[s] = *reinterpret_cast<const bf16x8*>(X + mrow * D + 16 * s + 8 * h);
#pragma unroll 1
  for (int ot = 0; ot < D / 32; ++ot) {
    const int o = ot * 32 + r;
    const float bo = b1[o];
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = bo;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bf16x8 wf = *reinterpret_cast<const bf16x8*>(W1 + (int64_t)o * D + 16 * s + 8 * h);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[s], wf, acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int mr = mfma_row(i, h);
      const uint16_t zb = f32_to_bf16_bits(fmaxf(acc[i], 0.f));
      zt[mr * LDZ + o] = zb;
      if (m0 + mr < rows) Zb[(m0 + mr) * D + o] = zb;
    }
  }
  __syncthreads();                                 // one wave: orders the LDS writes above before the reads below
  bf16x8 zf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) zf[s] = *reinterpret_cast<const bf16x8*>(zt + r * LDZ + 16 * s + 8 * h);
#pragma unroll 1
  for (int ot = 0; ot < D / 32; ++ot) {
    const int o = ot * 32 + r;
    const float bo = b2[o];
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = bo;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bf16x8 wf = *reinterpret_cast<const bf16x8*>(W2 + (int64_t)o * D + 16 * s + 8 * h);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zf[s], wf, acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t m = m0 + mfma_row(i, h);
      if (m < rows) Hb[m * D + o] = f32_to_bf16_bits(acc[i]);
    }
  }
}

extern "C" int cqlrec_encoder_fwd(const uint16_t* X_b, const uint16_t* W1_b, const float* b1, const uint16_t* W2_b,
                                  const float* b2, int64_t rows, int32_t d, uint16_t* Z_b, uint16_t* H_b,
                                  cqlrec_stream stream) {
  CQL_REQUIRE(X_b && W1_b && b1 && W2_b && b2 && Z_b && H_b, "encoder_fwd: NULL pointer");
  CQL_REQUIRE(d == 64 || d == 128 || d == 256, "encoder_fwd: d=%d unsupported", d);
  if (rows <= 0) return CQLREC_OK;
  dim3 grid(cql_ceil_div(rows, 32)), block(64);
  hipStream_t s = (hipStream_t)stream;
  CqlProfScope prof(CQLREC_PH_ENCODER_FWD, s);
#define ENCF_LAUNCH(DD) hipLaunchKernelGGL(encoder_fwd_kernel<DD>, grid, block, 0, s, X_b, W1_b, b1, W2_b, b2, rows, Z_b, H_b)
  if (d == 64) ENCF_LAUNCH(64); else if (d == 128) ENCF_LAUNCH(128); else ENCF_LAUNCH(256);
#undef ENCF_LAUNCH
  CQL_LAUNCH_CHECK("encoder_fwd");
  return CQLREC_OK;
}

extern "C" int cqlrec_linear_bf16(const uint16_t* X_b, const uint16_t* W_b, const float* bias, int64_t rows, int32_t d,
                                  int32_t relu, float* Y, uint16_t* Y_b, cqlrec_stream stream) {
  CQL_REQUIRE(X_b && W_b && bias, "linear_bf16: NULL pointer");
  CQL_REQUIRE(d == 64 || d == 128 || d == 256, "linear_bf16: d=%d unsupported", d);
  if (rows <= 0) return CQLREC_OK;
  dim3 grid(cql_ceil_div(rows, 32)), block(64);
  hipStream_t s = (hipStream_t)stream;
  CqlProfScope prof(CQLREC_PH_ENCODER_FWD, s);
#define LIN_LAUNCH(DD) hipLaunchKernelGGL(linear_bf16_kernel<DD>, grid, block, 0, s, X_b, W_b, bias, rows, relu, Y, Y_b)
  if (d == 64) LIN_LAUNCH(64); else if (d == 128) LIN_LAUNCH(128); else LIN_LAUNCH(256);
#undef LIN_LAUNCH
  CQL_LAUNCH_CHECK("linear_bf16");
  return CQLREC_OK;
}

// =============================================================================================================
// encoder backward in exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32: one f32 per lane for A and B, result
// bitwise a k-ordered fmaf chain -- cdna_hip_programming.md section 3 "FP32-input MFMA"); d x d weights, ~1 % of the step
//   dA1 = (dH W2_b) * [z_b > 0];  dh0 = dA1 W1_b;  gW2 = dH^T z_b;  gW1 = dA1^T h0_b;  gb2 = colsum dH;  gb1 = colsum dA1
// =============================================================================================================
// block = 32 rows (b) x all D columns: wave w owns columns [32w*.., ...) in tiles of 32; dH / dA1 tiles live in LDS
// (row stride D+1 floats: the A operand is read column-wise, 32 lanes x 32 rows).
template <int D>
__global__ __launch_bounds__(256) void enc_bwd_dx_kernel(const float* __restrict__ dH, const uint16_t* __restrict__ zb,
                                                         const uint16_t* __restrict__ W1b,
                                                         const uint16_t* __restrict__ W2b, int64_t rows,
                                                         float* __restrict__ dA1, float* __restrict__ dh0) {
  constexpr int LD = D + 1;
  constexpr int NT = D / 32;                 // column tiles
  __shared__ float tile[2][32 * LD];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * 32;
  for (int i = t; i < 32 * D; i += 256) {
    const int rr = i / D, cc = i % D;
    tile[0][rr * LD + cc] = (m0 + rr < rows) ? dH[(m0 + rr) * D + cc] : 0.f;
  }
  __syncthreads();
  // ---- dA1 tile = dH W2_b, masked by the relu ----
  for (int nt = wave; nt < NT; nt += 4) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 8
    for (int s = 0; s < D / 2; ++s) {
      const int o = 2 * s + h;
      const float a = tile[0][r * LD + o];                                     // A[b = r][o]
      const float bv = bf16_bits_to_f32(W2b[(int64_t)o * D + nt * 32 + r]);    // B[o][k = nt*32 + r]
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int rr = mfma_row(i, h), col = nt * 32 + r;
      const int64_t m = m0 + rr;
      float v = 0.f;
      if (m < rows) {
        const float z = bf16_bits_to_f32(zb[m * D + col]);
        v = (z > 0.f) ? acc[i] : 0.f;
        dA1[m * D + col] = v;
      }
      tile[1][rr * LD + col] = v;
    }
  }
  __syncthreads();
  // ---- dh0 tile = dA1 W1_b ----
  for (int nt = wave; nt < NT; nt += 4) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 8
    for (int s = 0; s < D / 2; ++s) {
      const int k = 2 * s + h;
      const float a = tile[1][r * LD + k];
      const float bv = bf16_bits_to_f32(W1b[(int64_t)k * D + nt * 32 + r]);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t m = m0 + mfma_row(i, h);
      if (m < rows) dh0[m * D + nt * 32 + r] = acc[i];
    }
  }
}

// partial dW over a chunk of ENC_CH rows: slab[pair][chunk][o][i] = sum_{b in chunk} G[b][o] X[b][i];
// one wave per 32x32 output tile and chunk: grid = (D/32 * D/32 / 4, nchunk, 2), 4 tiles per block.
#define ENC_CH 128
template <int D>
__global__ __launch_bounds__(256) void enc_bwd_dw_kernel(const float* __restrict__ dH, const float* __restrict__ dA1,
                                                         const uint16_t* __restrict__ zb,
                                                         const uint16_t* __restrict__ h0b, int64_t rows,
                                                         float* __restrict__ slab_w, float* __restrict__ slab_b,
                                                         int nchunk) {
  constexpr int NT = D / 32;
  const int pair = blockIdx.z, chunk = blockIdx.y;
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int tile_id = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int ot = tile_id / NT, it = tile_id % NT;
  const float* G = pair == 0 ? dH : dA1;
  const uint16_t* X = pair == 0 ? zb : h0b;
  const int64_t b0 = (int64_t)chunk * ENC_CH;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float bs = 0.f;
#pragma unroll 8
  for (int s = 0; s < ENC_CH / 2; ++s) {
    const int64_t m = b0 + 2 * s + h;
    float a = 0.f, bv = 0.f;
    if (m < rows) {
      a = G[m * D + ot * 32 + r];                            // A[o = r][b]  (row m of G: coalesced across r)
      bv = bf16_bits_to_f32(X[m * D + it * 32 + r]);         // B[b][i = r]
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
    bs += a;
  }
  float* sw = slab_w + ((int64_t)pair * nchunk + chunk) * D * D;
#pragma unroll
  for (int i = 0; i < 16; ++i) sw[(int64_t)(ot * 32 + mfma_row(i, h)) * D + it * 32 + r] = acc[i];
  if (it == 0) {
    bs += __shfl_xor(bs, 32);
    if (h == 0) slab_b[((int64_t)pair * nchunk + chunk) * D + ot * 32 + r] = bs;
  }
}

template <int D>
__global__ void enc_bwd_reduce_kernel(const float* __restrict__ slab_w, const float* __restrict__ slab_b, int nchunk,
                                      float* gW1, float* gb1, float* gW2, float* gb2) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int pair = blockIdx.y;
  float* gW = pair == 0 ? gW2 : gW1;
  float* gb = pair == 0 ? gb2 : gb1;
  if (idx < D * D) {
    float s = 0.f;
#pragma unroll 8
    for (int c = 0; c < nchunk; ++c) s += slab_w[((int64_t)pair * nchunk + c) * D * D + idx];
    gW[idx] = s;
  }
  if (idx < D) {
    float s = 0.f;
    for (int c = 0; c < nchunk; ++c) s += slab_b[((int64_t)pair * nchunk + c) * D + idx];
    gb[idx] = s;
  }
}

extern "C" int64_t cqlrec_encoder_bwd_ws_bytes(int64_t rows, int32_t d) {
  const int64_t nchunk = (rows + ENC_CH - 1) / ENC_CH;
  return (rows * d + 2 * nchunk * d * d + 2 * nchunk * d) * (int64_t)sizeof(float) + 256;
}

// parts: 1 = the input-side products (dA1 into ws, dh0: what the window-gather backward waits for), 2 = the weight / bias
// gradients (read dA1: behind part 1), 3 = both in order.  The step driver runs part 2 on another stream.
int cql_encoder_bwd_parts(const float* dH, const uint16_t* z_b, const uint16_t* h0_b, const uint16_t* W1_b,
                          const uint16_t* W2_b, int64_t rows, int32_t d, void* ws, int64_t ws_bytes, float* g_W1, float* g_b1,
                          float* g_W2, float* g_b2, float* dh0, int parts, hipStream_t s);
extern "C" int cqlrec_encoder_bwd(const float* dH, const uint16_t* z_b, const uint16_t* h0_b, const uint16_t* W1_b,
                                  const uint16_t* W2_b, int64_t rows, int32_t d, void* ws, int64_t ws_bytes,
                                  float* g_W1, float* g_b1, float* g_W2, float* g_b2, float* dh0,
                                  cqlrec_stream stream) {
  return cql_encoder_bwd_parts(dH, z_b, h0_b, W1_b, W2_b, rows, d, ws, ws_bytes, g_W1, g_b1, g_W2, g_b2, dh0, 3,
                               (hipStream_t)stream);
}
int cql_encoder_bwd_parts(const float* dH, const uint16_t* z_b, const uint16_t* h0_b, const uint16_t* W1_b,
                          const uint16_t* W2_b, int64_t rows, int32_t d, void* ws, int64_t ws_bytes, float* g_W1, float* g_b1,
                          float* g_W2, float* g_b2, float* dh0, int parts, hipStream_t stream) {
  CQL_REQUIRE(dH && z_b && h0_b && W1_b && W2_b && ws && g_W1 && g_b1 && g_W2 && g_b2 && dh0, "encoder_bwd: NULL pointer");
  CQL_REQUIRE(d == 64 || d == 128 || d == 256, "encoder_bwd: d=%d unsupported", d);
  CQL_REQUIRE(rows > 0, "encoder_bwd: rows=%lld", (long long)rows);
  CQL_REQUIRE(ws_bytes >= cqlrec_encoder_bwd_ws_bytes(rows, d), "encoder_bwd: workspace too small");
  const int nchunk = cql_ceil_div(rows, ENC_CH);
  float* dA1 = (float*)ws;
  float* slab_w = dA1 + rows * d;
  float* slab_b = slab_w + (int64_t)2 * nchunk * d * d;
  hipStream_t s = (hipStream_t)stream;
  CqlProfScope prof(CQLREC_PH_ENCODER_BWD, s);
#define ENC_LAUNCH(DD)                                                                                               \
  do {                                                                                                               \
    if (parts & 1)                                                                                                   \
      hipLaunchKernelGGL(enc_bwd_dx_kernel<DD>, dim3(cql_ceil_div(rows, 32)), dim3(256), 0, s, dH, z_b, W1_b, W2_b, \
                         rows, dA1, dh0);                                                                            \
    if (parts & 2) {                                                                                                 \
      hipLaunchKernelGGL(enc_bwd_dw_kernel<DD>, dim3((DD / 32) * (DD / 32) / 4, nchunk, 2), dim3(256), 0, s, dH, dA1, z_b, \
                         h0_b, rows, slab_w, slab_b, nchunk);                                                        \
      hipLaunchKernelGGL(enc_bwd_reduce_kernel<DD>, dim3(cql_ceil_div(DD * DD, 256), 2), dim3(256), 0, s, slab_w,    \
                         slab_b, nchunk, g_W1, g_b1, g_W2, g_b2);                                                    \
    }                                                                                                                \
  } while (0)
  if (d == 64) ENC_LAUNCH(64); else if (d == 128) ENC_LAUNCH(128); else ENC_LAUNCH(256);
#undef ENC_LAUNCH
  CQL_LAUNCH_CHECK("encoder_bwd");
  return CQLREC_OK;
}

// =============================================================================================================
// gather-dot: out[r] = <H_b[r], E_b[idx[r]]> + b[idx[r]]
// =============================================================================================================
template <int D>
__global__ __launch_bounds__(256) void gather_dot_kernel(const uint16_t* __restrict__ H, const uint16_t* __restrict__ E,
                                                         const float* __restrict__ b, const int32_t* __restrict__ idx,
                                                         int64_t rows, float* __restrict__ out) {
  constexpr int LPR = D / 8, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r = ((int64_t)blockIdx.x * 4 + wave) * RPW + lane / LPR;
  const int c = lane % LPR;
  float s = 0.f;
  int j = 0;
  if (r < rows) {
    j = idx[r];
    const uint4 hv = *reinterpret_cast<const uint4*>(H + r * D + c * 8);
    const uint4 ev = *reinterpret_cast<const uint4*>(E + (int64_t)j * D + c * 8);
    float hf[8], ef[8];
    unpack_bf16x8(hv, hf);
    unpack_bf16x8(ev, ef);
#pragma unroll
    for (int i = 0; i < 8; ++i) s = fmaf(hf[i], ef[i], s);
  }
#pragma unroll
  for (int off = 1; off < LPR; off <<= 1) s += __shfl_xor(s, off);
  if (r < rows && c == 0) out[r] = s + b[j];
}

extern "C" int cqlrec_gather_dot(const uint16_t* H_b, const uint16_t* E_b, const float* b, const int32_t* idx,
                                 int64_t rows, int32_t d, float* out, cqlrec_stream stream) {
  CQL_REQUIRE(H_b && E_b && b && idx && out, "gather_dot: NULL pointer");
  CQL_REQUIRE(d == 64 || d == 128 || d == 256, "gather_dot: d=%d unsupported", d);
  if (rows <= 0) return CQLREC_OK;
  hipStream_t s = (hipStream_t)stream;
  CqlProfScope prof(CQLREC_PH_QHEAD_SMALL, s);
  const int rpb = 4 * 64 / (d / 8);
  dim3 grid(cql_ceil_div(rows, rpb)), block(256);
#define GD_LAUNCH(DD) hipLaunchKernelGGL(gather_dot_kernel<DD>, grid, block, 0, s, H_b, E_b, b, idx, rows, out)
  if (d == 64) GD_LAUNCH(64); else if (d == 128) GD_LAUNCH(128); else GD_LAUNCH(256);
#undef GD_LAUNCH
  CQL_LAUNCH_CHECK("gather_dot");
  return CQLREC_OK;
}

// =============================================================================================================
// TD target + CQL loss (single block; deterministic tree reduction)
// =============================================================================================================
__global__ __launch_bounds__(256) void td_loss_kernel(const float* __restrict__ q_a, const float* __restrict__ lse,
                                                      const float* __restrict__ q_targ, const float* __restrict__ rew,
                                                      const float* __restrict__ done, int batch, float gamma,
                                                      float alpha, float inv_batch, float* __restrict__ coef,
                                                      float* __restrict__ y, float* __restrict__ loss_out) {
  __shared__ float red[256];
  float s = 0.f;
  for (int b = threadIdx.x; b < batch; b += 256) {
    const float yy = rew[b] + gamma * (1.0f - done[b]) * q_targ[b];
    const float delta = q_a[b] - yy;
    s += 0.5f * delta * delta + alpha * (lse[b] - q_a[b]);
    coef[b] = (delta - alpha) * inv_batch;
    if (y) y[b] = yy;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0 && loss_out) *loss_out = red[0] * inv_batch;
}

// The same in two launches for the step driver: the dQ coefficients -- all the backward needs -- by as many blocks as
// there are rows to spread, and the loss VALUE (which nothing in the step waits for) summed by one block from the terms
// the first launch leaves, in td_loss_kernel's order: coefficients, targets and loss bit-identical to the one launch.
__global__ __launch_bounds__(256) void td_coef_kernel(const float* __restrict__ q_a, const float* __restrict__ lse,
                                                      const float* __restrict__ q_targ, const float* __restrict__ rew,
                                                      const float* __restrict__ done, int batch, float gamma, float alpha,
                                                      float inv_batch, float* __restrict__ coef, float* __restrict__ y,
                                                      float* __restrict__ term) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= batch) return;
  const float yy = rew[b] + gamma * (1.0f - done[b]) * q_targ[b];
  const float delta = q_a[b] - yy;
  term[b] = 0.5f * delta * delta + alpha * (lse[b] - q_a[b]);
  coef[b] = (delta - alpha) * inv_batch;
  if (y) y[b] = yy;
}
__global__ __launch_bounds__(256) void td_loss_sum_kernel(const float* __restrict__ term, int batch, float inv_batch,
                                                          float* __restrict__ loss_out) {
  __shared__ float red[256];
  float s = 0.f;
  for (int b = threadIdx.x; b < batch; b += 256) s += term[b];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss_out = red[0] * inv_batch;
}
int cql_td_coef(const float* q_a, const float* lse, const float* q_targ, const float* rew, const float* done, int32_t batch,
                float gamma, float alpha, float inv_batch, float* coef, float* y, float* term, hipStream_t s) {
  CQL_REQUIRE(q_a && lse && q_targ && rew && done && coef && term, "td_coef: NULL pointer");
  CQL_REQUIRE(batch > 0, "td_coef: batch=%d", batch);
  CqlProfScope prof(CQLREC_PH_QHEAD_SMALL, s);
  hipLaunchKernelGGL(td_coef_kernel, dim3((unsigned)cql_ceil_div(batch, 256)), dim3(256), 0, s, q_a, lse, q_targ, rew, done,
                     batch, gamma, alpha, inv_batch, coef, y, term);
  CQL_LAUNCH_CHECK("td_coef");
  return CQLREC_OK;
}
int cql_td_loss_sum(const float* term, int32_t batch, float inv_batch, float* loss_out, hipStream_t s) {
  CQL_REQUIRE(term && loss_out && batch > 0, "td_loss_sum: bad arguments");
  CqlProfScope prof(CQLREC_PH_QHEAD_SMALL, s);
  hipLaunchKernelGGL(td_loss_sum_kernel, dim3(1), dim3(256), 0, s, term, batch, inv_batch, loss_out);
  CQL_LAUNCH_CHECK("td_loss_sum");
  return CQLREC_OK;
}

extern "C" int cqlrec_td_loss(const float* q_a, const float* lse, const float* q_targ, const float* rew,
                              const float* done, int32_t batch, float gamma, float alpha, float inv_batch, float* coef,
                              float* y, float* loss_out, cqlrec_stream stream) {
  CQL_REQUIRE(q_a && lse && q_targ && rew && done && coef, "td_loss: NULL pointer");
  CQL_REQUIRE(batch > 0, "td_loss: batch=%d", batch);
  CqlProfScope prof(CQLREC_PH_QHEAD_SMALL, (hipStream_t)stream);
  hipLaunchKernelGGL(td_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, q_a, lse, q_targ, rew, done, batch,
                     gamma, alpha, inv_batch, coef, y, loss_out);
  CQL_LAUNCH_CHECK("td_loss");
  return CQLREC_OK;
}

// =============================================================================================================
// fused Adam + Polyak + bf16 shadows.  Pure HBM streaming: 20 B read + 24 B written per parameter.
// Compiled with -ffp-contract=off so the expression order below is the normative one (oracle.adam_ema_step).
// =============================================================================================================
template <bool NT>
__global__ __launch_bounds__(256) void adam_ema_kernel(float4* __restrict__ theta, float4* __restrict__ grads,
                                                       float4* __restrict__ m, float4* __restrict__ v,
                                                       float4* __restrict__ target, uint2* __restrict__ theta_b,
                                                       uint2* __restrict__ target_b, int64_t n4, float step_size,
                                                       float sqrt_bc2, float beta1, float beta2, float eps, float tau,
                                                       int zero_grads) {
  const float omb1 = 1.0f - beta1, omb2 = 1.0f - beta2, omt = 1.0f - tau;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 g4 = ld4<NT>(grads + i);
    float4 p4 = ld4<NT>(theta + i), m4 = ld4<NT>(m + i), v4 = ld4<NT>(v + i), t4 = ld4<NT>(target + i);
    const float g[4] = {g4.x, g4.y, g4.z, g4.w};
    float p[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w},
          tt[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      adam_ema_elem(g[k], p[k], mm[k], vv[k], tt[k], step_size, sqrt_bc2, beta1, beta2, eps, tau, omb1, omb2, omt);
    }
    st4<NT>(theta + i, p[0], p[1], p[2], p[3]);
    st4<NT>(m + i, mm[0], mm[1], mm[2], mm[3]);
    st4<NT>(v + i, vv[0], vv[1], vv[2], vv[3]);
    st4<NT>(target + i, tt[0], tt[1], tt[2], tt[3]);
    theta_b[i] = make_uint2(pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]));
    target_b[i] = make_uint2(pack_bf16x2(tt[0], tt[1]), pack_bf16x2(tt[2], tt[3]));
    if (zero_grads) st4<NT>(grads + i, 0.f, 0.f, 0.f, 0.f);
  }
}

// what qde_fixup_kernel would have added to the four gradient elements 4 i4 .. 4 i4 + 3 (relative to the start of the
// updated range).  D and the items per group are powers of two and every offset is a multiple of 4, so the four elements
// belong to one item (rows) or to four items of one group (column sums); 32-bit arithmetic (the host checks the range).
__device__ __forceinline__ void adam_fix4(const CqlAdamFix& f, float (&g)[4], int64_t i4, int ld, int li) {
  const int64_t e = 4 * i4;
  uint32_t item;
  int col = -1;
  if (e >= f.rows_off && e < f.rows_off + f.n_items * f.D) {
    const uint64_t r = (uint64_t)(e - f.rows_off);
    item = (uint32_t)(r >> ld);
    col = (int)(r & (uint64_t)(f.D - 1));
  } else if (e >= f.cs_off && e < f.cs_off + f.n_items) {
    item = (uint32_t)(e - f.cs_off);
  } else {
    return;
  }
#ifdef ADAMFIX_ABL_NOLOOP      // timing-only build: the gradient without the slabs
  return;
#endif
  const uint32_t grp = item >> li, row = item & (uint32_t)(f.items - 1);
  const uint32_t W = (uint32_t)f.G * (uint32_t)f.T, nblk = (uint32_t)f.nblk, lo = grp * (uint32_t)f.T, hi = lo + (uint32_t)f.T;
  for (uint32_t p = ((lo + 1) * nblk + W - 1) / W; p < nblk; ++p) {     // the pieces, in block order
    const uint32_t u0 = (uint32_t)((uint64_t)p * W / nblk);
    if (u0 >= hi) break;
    if (u0 <= lo) continue;
    if ((uint32_t)(((uint64_t)p + 1) * W / nblk) == u0) continue;
    if (col >= 0) {
      const float4 s4 = *reinterpret_cast<const float4*>(f.slab + ((int64_t)p * f.items + row) * f.D + col);
      g[0] += f.scale * s4.x; g[1] += f.scale * s4.y; g[2] += f.scale * s4.z; g[3] += f.scale * s4.w;
    } else {
      const float* sc = f.slab_cs + (int64_t)p * f.items + row;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if ((int64_t)item + k < f.n_items) g[k] += f.scale * sc[k];
    }
  }
}

template <bool NT>
__global__ __launch_bounds__(256) void adam_ema_fix_kernel(float4* __restrict__ theta, float4* __restrict__ grads,
                                                           float4* __restrict__ m, float4* __restrict__ v,
                                                           float4* __restrict__ target, uint2* __restrict__ theta_b,
                                                           uint2* __restrict__ target_b, int64_t n4, float step_size,
                                                           float sqrt_bc2, float beta1, float beta2, float eps, float tau,
                                                           int zero_grads, CqlAdamFix fix) {
  const float omb1 = 1.0f - beta1, omb2 = 1.0f - beta2, omt = 1.0f - tau;
  const int ld = __builtin_ctz((unsigned)fix.D), li = __builtin_ctz((unsigned)fix.items);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 g4 = ld4<NT>(grads + i);
    float4 p4 = ld4<NT>(theta + i), m4 = ld4<NT>(m + i), v4 = ld4<NT>(v + i), t4 = ld4<NT>(target + i);
    float g[4] = {g4.x, g4.y, g4.z, g4.w};
    adam_fix4(fix, g, i, ld, li);
    float p[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w},
          tt[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      adam_ema_elem(g[k], p[k], mm[k], vv[k], tt[k], step_size, sqrt_bc2, beta1, beta2, eps, tau, omb1, omb2, omt);
    }
    st4<NT>(theta + i, p[0], p[1], p[2], p[3]);
    st4<NT>(m + i, mm[0], mm[1], mm[2], mm[3]);
    st4<NT>(v + i, vv[0], vv[1], vv[2], vv[3]);
    st4<NT>(target + i, tt[0], tt[1], tt[2], tt[3]);
    theta_b[i] = make_uint2(pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]));
    target_b[i] = make_uint2(pack_bf16x2(tt[0], tt[1]), pack_bf16x2(tt[2], tt[3]));
    if (zero_grads) st4<NT>(grads + i, 0.f, 0.f, 0.f, 0.f);
  }
}

int cql_adam_ema_fix(float* theta, float* grads, float* m, float* v, float* target, uint16_t* theta_b, uint16_t* target_b,
                     int64_t n, float step_size, float sqrt_bc2, float beta1, float beta2, float eps, float tau,
                     int32_t zero_grads, const CqlAdamFix* fix, hipStream_t stream) {
  const bool pow2 = fix && fix->valid && (fix->D & (fix->D - 1)) == 0 && (fix->items & (fix->items - 1)) == 0 &&
                    fix->rows_off % 4 == 0 && fix->cs_off % 4 == 0 &&
                    (int64_t)fix->G * fix->T * ((int64_t)fix->nblk + 1) < (1ll << 31) && fix->n_items < (1ll << 31);
  CQL_REQUIRE(!fix || !fix->valid || pow2, "adam_ema: deferred fix-up outside the supported range");
  if (!fix || !fix->valid)
    return cqlrec_adam_ema(theta, grads, m, v, target, theta_b, target_b, n, step_size, sqrt_bc2, beta1, beta2, eps, tau,
                           zero_grads, (cqlrec_stream)stream);
  CQL_REQUIRE(theta && grads && m && v && target && theta_b && target_b, "adam_ema: NULL pointer");
  CQL_REQUIRE(n > 0 && n % 4 == 0, "adam_ema: n=%lld must be a positive multiple of 4", (long long)n);
  const int64_t n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  CqlProfScope prof(CQLREC_PH_ADAM, stream);
  hipLaunchKernelGGL(adam_ema_fix_kernel<true>, dim3(blocks), dim3(256), 0, stream, (float4*)theta, (float4*)grads,
                     (float4*)m, (float4*)v, (float4*)target, (uint2*)theta_b, (uint2*)target_b, n4, step_size, sqrt_bc2,
                     beta1, beta2, eps, tau, zero_grads, *fix);
  CQL_LAUNCH_CHECK("adam_ema (deferred fix-up)");
  return CQLREC_OK;
}

extern "C" int cqlrec_adam_ema(float* theta, float* grads, float* m, float* v, float* target, uint16_t* theta_b,
                               uint16_t* target_b, int64_t n, float step_size, float sqrt_bc2, float beta1, float beta2,
                               float eps, float tau, int32_t zero_grads, cqlrec_stream stream) {
  CQL_REQUIRE(theta && grads && m && v && target && theta_b && target_b, "adam_ema: NULL pointer");
  CQL_REQUIRE(n > 0 && n % 4 == 0, "adam_ema: n=%lld must be a positive multiple of 4", (long long)n);
  const int64_t n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  static int nt = -1;
  if (nt < 0) {
    const char* e = getenv("CQL_ADAM_NT");
    nt = (e && *e == '0') ? 0 : 1;
  }
  CqlProfScope prof(CQLREC_PH_ADAM, (hipStream_t)stream);
  if (nt)
    hipLaunchKernelGGL(adam_ema_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float4*)theta,
                       (float4*)grads, (float4*)m, (float4*)v, (float4*)target, (uint2*)theta_b, (uint2*)target_b, n4,
                       step_size, sqrt_bc2, beta1, beta2, eps, tau, zero_grads);
  else
    hipLaunchKernelGGL(adam_ema_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float4*)theta,
                       (float4*)grads, (float4*)m, (float4*)v, (float4*)target, (uint2*)theta_b, (uint2*)target_b, n4,
                       step_size, sqrt_bc2, beta1, beta2, eps, tau, zero_grads);
  CQL_LAUNCH_CHECK("adam_ema");
  return CQLREC_OK;
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float4* __restrict__ src, uint2* __restrict__ dst,
                                                        int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 p = src[i];
    dst[i] = make_uint2(pack_bf16x2(p.x, p.y), pack_bf16x2(p.z, p.w));
  }
}
extern "C" int cqlrec_cast_bf16(const float* src, uint16_t* dst_b, int64_t n, cqlrec_stream stream) {
  CQL_REQUIRE(src && dst_b, "cast_bf16: NULL pointer");
  CQL_REQUIRE(n > 0 && n % 4 == 0, "cast_bf16: n=%lld must be a positive multiple of 4", (long long)n);
  const int64_t n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)src,
                     (uint2*)dst_b, n4);
  CQL_LAUNCH_CHECK("cast_bf16");
  return CQLREC_OK;
}
