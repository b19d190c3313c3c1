// Calibration probe: cycles per v_mfma_f32_32x32x16_bf16 in the instruction mixes the Q-head kernels use, ONE wave per
// SIMD (4 waves per block, one block per CU) and two.  s_memtime around a loop of REP iterations, median over blocks.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_probe.hip -o gpurun_out/mfma_probe && gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define REP 256

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* out, float* sink, const float* src) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[i];
  __syncthreads();
  bf16x8 a[8], b[8];
  for (int s = 0; s < 8; ++s)
    for (int j = 0; j < 8; ++j) {
      a[s][j] = (__bf16)src[(lane * 8 + j + s * 37) & 4095];
      b[s][j] = (__bf16)src[(lane * 8 + j + s * 91 + 7) & 4095];
    }
  f32x16 acc[8];
  for (int k = 0; k < 8; ++k)
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  float v[16], c = 0.f;
  for (int i = 0; i < 16; ++i) v[i] = src[(lane + i) & 4095];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int it = 0; it < REP; ++it) {
    if constexpr (MODE == 0) {          // one dependent chain of 8
#pragma unroll
      for (int s = 0; s < 8; ++s) { acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[s], acc[0], 0, 0, 0); FENCE(); }
    } else if constexpr (MODE == 1) {   // 8 independent accumulators
#pragma unroll
      for (int s = 0; s < 8; ++s) { acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[s], acc[s], 0, 0, 0); FENCE(); }
    } else if constexpr (MODE == 2) {   // chain + 7 VALU (2 fma, 2 exp, 2 add, 1 cvt) behind each MFMA, independent data
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[s], acc[0], 0, 0, 0);
        FENCE();
        float t0_, t1_; unsigned w;
        asm volatile("v_fmamk_f32 %0, %4, 0x3fb8aa3b, %6\n\tv_fmamk_f32 %1, %5, 0x3fb8aa3b, %6\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\t"
                     "v_add_f32 %3, %3, %0\n\tv_add_f32 %3, %3, %1\n\tv_cvt_pk_bf16_f32 %2, %0, %1"
                     : "=&v"(t0_), "=&v"(t1_), "=&v"(w), "+v"(c) : "v"(v[2 * s]), "v"(v[2 * s + 1]), "v"(v[(s + 3) & 15]));
        v[(s + 5) & 15] = __uint_as_float(w);
        FENCE();
      }
    } else if constexpr (MODE == 3) {   // independent accumulators + the same VALU
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[s], acc[s], 0, 0, 0);
        FENCE();
        float t0_, t1_; unsigned w;
        asm volatile("v_fmamk_f32 %0, %4, 0x3fb8aa3b, %6\n\tv_fmamk_f32 %1, %5, 0x3fb8aa3b, %6\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\t"
                     "v_add_f32 %3, %3, %0\n\tv_add_f32 %3, %3, %1\n\tv_cvt_pk_bf16_f32 %2, %0, %1"
                     : "=&v"(t0_), "=&v"(t1_), "=&v"(w), "+v"(c) : "v"(v[2 * s]), "v"(v[2 * s + 1]), "v"(v[(s + 3) & 15]));
        v[(s + 5) & 15] = __uint_as_float(w);
        FENCE();
      }
    } else if constexpr (MODE == 4) {   // chain + 3 plain VALU (adds) behind each MFMA
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[s], acc[0], 0, 0, 0);
        FENCE();
        asm volatile("v_add_f32 %0, %0, %3\n\tv_add_f32 %1, %1, %3\n\tv_add_f32 %2, %2, %3" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]) : "v"(v[8]));
        FENCE();
      }
    } else if constexpr (MODE == 5) {   // chain + 2 ds_read_b128 behind each MFMA (data unused until the end)
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[s], acc[0], 0, 0, 0);
        FENCE();
        typedef __attribute__((ext_vector_type(4))) float f4;
        const f4 x = *(const f4*)&lds[((lane * 4 + s * 256 + it * 4) & 4092)];
        const f4 y = *(const f4*)&lds[((lane * 4 + s * 256 + 2048 + it * 4) & 4092)];
        v[s] += x[0] + y[1];
        FENCE();
      }
    } else if constexpr (MODE >= 8 && MODE <= 12) {   // LDS reads issued between MFMAs, data NOT consumed in the loop
      typedef __attribute__((ext_vector_type(4))) float f4;
      typedef __attribute__((address_space(3))) f4 lf4;
      const lf4* base = (const lf4*)((__attribute__((address_space(3))) float*)lds) + (lane ^ (it & 7));
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[s], acc[0], 0, 0, 0);
        FENCE();
        constexpr int NR = (MODE == 8 || MODE == 12) ? 1 : (MODE == 9) ? 2 : (MODE == 10) ? 3 : 4;
        f4 x[NR];
#pragma unroll
        for (int e = 0; e < NR; ++e) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[e]) : "v"(base), "i"((s * 4 + e) * 1024 % 16384));
        FENCE();
        if constexpr (MODE == 12) {
          float t0_, t1_; unsigned w;
          asm volatile("v_fmamk_f32 %0, %4, 0x3fb8aa3b, %6\n\tv_fmamk_f32 %1, %5, 0x3fb8aa3b, %6\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\t"
                       "v_add_f32 %3, %3, %0\n\tv_add_f32 %3, %3, %1\n\tv_cvt_pk_bf16_f32 %2, %0, %1"
                       : "=&v"(t0_), "=&v"(t1_), "=&v"(w), "+v"(c) : "v"(v[2 * s]), "v"(v[2 * s + 1]), "v"(v[(s + 3) & 15]));
          v[(s + 5) & 15] = __uint_as_float(w);
          FENCE();
        }
#pragma unroll
        for (int e = 0; e < NR; ++e) asm volatile("" :: "v"(x[e]));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (MODE == 13) {   // half-chunks: {2 fmamk, 1 exp} / {1 exp, 2 add, 1 cvt} behind alternate MFMAs
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[s], acc[0], 0, 0, 0);
        FENCE();
        if ((s & 1) == 0) {
          asm volatile("v_fmamk_f32 %0, %2, 0x3fb8aa3b, %4\n\tv_fmamk_f32 %1, %3, 0x3fb8aa3b, %4\n\tv_exp_f32 %0, %0"
                       : "=&v"(v[12]), "=&v"(v[13]) : "v"(v[2 * (s >> 1)]), "v"(v[2 * (s >> 1) + 1]), "v"(v[11]));
        } else {
          unsigned w;
          asm volatile("v_exp_f32 %1, %1\n\tv_add_f32 %3, %3, %0\n\ts_nop 0\n\tv_add_f32 %3, %3, %1\n\tv_cvt_pk_bf16_f32 %2, %0, %1"
                       : "+v"(v[12]), "+v"(v[13]), "=&v"(w), "+v"(c));
          v[8 + (s >> 1)] = __uint_as_float(w);
        }
        FENCE();
      }
    } else if constexpr (MODE == 6) {   // the MFMA consumes the previous VALU result as B operand (pack -> MFMA)
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        unsigned w0, w1, w2, w3;
        asm volatile("v_cvt_pk_bf16_f32 %0, %4, %5\n\tv_cvt_pk_bf16_f32 %1, %5, %6\n\tv_cvt_pk_bf16_f32 %2, %6, %7\n\tv_cvt_pk_bf16_f32 %3, %7, %4"
                     : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3) : "v"(v[s]), "v"(v[s + 1]), "v"(v[s + 2]), "v"(v[s + 3]));
        typedef __attribute__((ext_vector_type(4))) unsigned u4;
        u4 u = {w0, w1, w2, w3};
        acc[s & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], __builtin_bit_cast(bf16x8, u), acc[s & 3], 0, 0, 0);
        FENCE();
      }
    } else if constexpr (MODE == 14 || MODE == 15) {   // v_mfma_f32_16x16x32_bf16 (half the flops of a 32x32x16): chain / independent
      typedef __attribute__((ext_vector_type(4))) float f32x4_;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        f32x4_ c4 = {acc[MODE == 14 ? 0 : s][0], acc[MODE == 14 ? 0 : s][1], acc[MODE == 14 ? 0 : s][2], acc[MODE == 14 ? 0 : s][3]};
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s], b[s], c4, 0, 0, 0);
        acc[MODE == 14 ? 0 : s][0] = c4[0]; acc[MODE == 14 ? 0 : s][1] = c4[1];
        acc[MODE == 14 ? 0 : s][2] = c4[2]; acc[MODE == 14 ? 0 : s][3] = c4[3];
        FENCE();
      }
    } else if constexpr (MODE >= 16 && MODE <= 19) {   // VALU rates without any MFMA: 8 instructions per "slot"
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if constexpr (MODE == 16) asm volatile("v_exp_f32 %0, %0" : "+v"(v[s]));
        else if constexpr (MODE == 17) asm volatile("v_fmamk_f32 %0, %0, 0x3fb8aa3b, %1" : "+v"(v[s]) : "v"(v[15]));
        else if constexpr (MODE == 18) {
          typedef __attribute__((ext_vector_type(2))) float f32x2_;
          f32x2_ x = {v[2 * (s & 3)], v[2 * (s & 3) + 1]}, y2 = {v[8], v[9]}, z2 = {v[10], v[11]};
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y2), "v"(z2));
          v[2 * (s & 3)] = x[0]; v[2 * (s & 3) + 1] = x[1];
        } else {
          unsigned w;
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(v[s]), "v"(v[s + 8]));
          v[s] = __uint_as_float(w);
        }
        FENCE();
      }
    } else if constexpr (MODE == 7) {   // VALU reads the accumulator of a chain that ended 3 MFMAs ago (acc -> exp -> pack)
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        acc[s & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[s], acc[s & 1], 0, 0, 0);
        FENCE();
        float t0_, t1_; unsigned w;
        asm volatile("v_fmamk_f32 %0, %4, 0x3fb8aa3b, %6\n\tv_fmamk_f32 %1, %5, 0x3fb8aa3b, %6\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\t"
                     "v_add_f32 %3, %3, %0\n\tv_add_f32 %3, %3, %1\n\tv_cvt_pk_bf16_f32 %2, %0, %1"
                     : "=&v"(t0_), "=&v"(t1_), "=&v"(w), "+v"(c) : "v"(acc[2][2 * s]), "v"(acc[2][2 * s + 1]), "v"(v[3]));
        v[(s + 5) & 15] = __uint_as_float(w);
        FENCE();
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  float r = c;
  for (int k = 0; k < 8; ++k) for (int i = 0; i < 16; ++i) r += acc[k][i];
  for (int i = 0; i < 16; ++i) r += v[i];
  if (r == 1234.5f) sink[0] = r;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* what, int blocks_per_cu, unsigned long long* d_out, float* d_sink, float* d_src) {
  const int nb = 256 * blocks_per_cu;
  std::vector<unsigned long long> h(nb);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(nb), dim3(256), 0, 0, d_out, d_sink, d_src);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), d_out, nb * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("%-64s waves/SIMD=%d  cycles per MFMA: median %.1f  (min %.1f max %.1f)\n", what, blocks_per_cu,
         h[nb / 2] / (double)(REP * 8), h[0] / (double)(REP * 8), h[nb - 1] / (double)(REP * 8));
}

int main() {
  unsigned long long* d_out; float *d_sink, *d_src;
  hipMalloc(&d_out, 1024 * 8); hipMalloc(&d_sink, 64); hipMalloc(&d_src, 4096 * 4);
  std::vector<float> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
  hipMemcpy(d_src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  for (int w = 1; w <= 2; ++w) {
    run<0>("dependent chain of 8", w, d_out, d_sink, d_src);
    run<1>("8 independent accumulators", w, d_out, d_sink, d_src);
    run<2>("chain + chunk (2 fmamk, 2 exp, 2 add, 1 cvt) per MFMA", w, d_out, d_sink, d_src);
    run<3>("independent + chunk per MFMA", w, d_out, d_sink, d_src);
    run<4>("chain + 3 v_add per MFMA", w, d_out, d_sink, d_src);
    run<5>("chain + 2 ds_read_b128 per MFMA", w, d_out, d_sink, d_src);
    run<6>("4 cvt_pk -> MFMA B operand", w, d_out, d_sink, d_src);
    run<7>("2 alternating chains + chunk reading a third accumulator", w, d_out, d_sink, d_src);
    run<8>("chain + 1 ds_read_b128 per MFMA (not consumed)", w, d_out, d_sink, d_src);
    run<9>("chain + 2 ds_read_b128 per MFMA (not consumed)", w, d_out, d_sink, d_src);
    run<10>("chain + 3 ds_read_b128 per MFMA (not consumed)", w, d_out, d_sink, d_src);
    run<11>("chain + 4 ds_read_b128 per MFMA (not consumed)", w, d_out, d_sink, d_src);
    run<12>("chain + chunk + 1 ds_read_b128 per MFMA", w, d_out, d_sink, d_src);
    run<13>("chain + half-chunks (3 / 4 VALU) behind alternate MFMAs", w, d_out, d_sink, d_src);
    run<14>("16x16x32 bf16, dependent chain (half the flops per MFMA)", w, d_out, d_sink, d_src);
    run<15>("16x16x32 bf16, 8 independent accumulators", w, d_out, d_sink, d_src);
    run<16>("no MFMA: v_exp_f32 (cycles per instruction)", w, d_out, d_sink, d_src);
    run<17>("no MFMA: v_fmamk_f32", w, d_out, d_sink, d_src);
    run<18>("no MFMA: v_pk_fma_f32 (two fp32 FMAs)", w, d_out, d_sink, d_src);
    run<19>("no MFMA: v_cvt_pk_bf16_f32", w, d_out, d_sink, d_src);
  }
  return 0;
}
