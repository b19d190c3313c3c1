// The Adam + Polyak arithmetic of one parameter, shared by the streaming optimizer kernels (misc.hip) and by the fused
// epilogue of the item-side backward (qhead_de2.hip).  The expression order is the normative one (oracle.adam_ema_step):
// no fma contraction, IEEE division and square root -- whichever kernel updates a parameter, the bits are the same.
#pragma once
#include "common.h"

struct AdamK {
  float step_size, sqrt_bc2, beta1, beta2, eps, tau;
};

__device__ __forceinline__ void adam_ema_elem(float g, float& p, float& m, float& v, float& t, float step_size, float sqrt_bc2,
                                              float beta1, float beta2, float eps, float tau, float omb1, float omb2,
                                              float omt) {
#pragma clang fp contract(off)
  m = beta1 * m + omb1 * g;
  v = beta2 * v + (omb2 * g) * g;
  const float denom = sqrtf(v) / sqrt_bc2 + eps;
  p = p - step_size * (m / denom);
  t = omt * t + tau * p;
}

// NT: the fp32 streams (read once, written once per step) bypass the caches with non-temporal accesses, so that the
// optimizer does not evict the bf16 E_out shadow -- which the Q-head kernels keep re-reading from L2 / Infinity Cache --
// when it runs next to them; the bf16 shadows it writes stay cacheable (they are what the next kernels read).
template <bool NT>
__device__ __forceinline__ float4 ld4(const float4* p) {
  if constexpr (NT) {
    const float* f = reinterpret_cast<const float*>(p);
    return make_float4(__builtin_nontemporal_load(f), __builtin_nontemporal_load(f + 1), __builtin_nontemporal_load(f + 2),
                       __builtin_nontemporal_load(f + 3));
  } else {
    return *p;
  }
}
template <bool NT>
__device__ __forceinline__ void st4(float4* p, float a, float b, float c, float d) {
  if constexpr (NT) {
    float* f = reinterpret_cast<float*>(p);
    __builtin_nontemporal_store(a, f);
    __builtin_nontemporal_store(b, f + 1);
    __builtin_nontemporal_store(c, f + 2);
    __builtin_nontemporal_store(d, f + 3);
  } else {
    *p = make_float4(a, b, c, d);
  }
}
