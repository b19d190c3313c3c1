// Which pairs of streams block each other's DISPATCH?  A kernel whose grid is far larger than what fits on the chip keeps
// its queue's dispatcher busy for its whole duration; a one-block kernel on another stream launched right behind it either
// starts at once (different dispatch pipe) or only when the big grid has been handed out (same pipe / same queue).
// r3: this -- not queue sharing -- is what made the training step lose its concurrency depending on the ORDER in which the
// process had created its streams (MI355X, ROCm 7.2: streams appear to get their hardware queues on the 4 compute pipes
// round-robin in creation order).    pipe_probe [n_streams=10]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ __launch_bounds__(64) void hog(int spins) {
  extern __shared__ volatile char pad[];  // 60 KiB of dynamic LDS: two blocks per CU, 512 resident of a 12288-block grid
  pad[threadIdx.x] = 1;
  for (int i = 0; i < spins; ++i) __builtin_amdgcn_s_sleep(127);
  pad[threadIdx.x + 64] = pad[threadIdx.x];
}
__global__ void tiny() {}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 10;
  tiny<<<1, 64>>>();              // the default stream's queue exists first
  hipDeviceSynchronize();
  std::vector<hipStream_t> s(n);
  for (auto& x : s) {
    hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    tiny<<<1, 64, 0, x>>>();      // bind its queue now, in creation order
  }
  hipDeviceSynchronize();
  hipEvent_t a0, a1, b1;
  hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b1);
  auto conflict = [&](hipStream_t x, hipStream_t y, float* t_hog, float* t_tiny) {
    hipDeviceSynchronize();
    hipEventRecord(a0, x);
    hog<<<12288, 64, 60 * 1024, x>>>(4);
    hipEventRecord(a1, x);
    tiny<<<1, 64, 0, y>>>();
    hipEventRecord(b1, y);
    hipDeviceSynchronize();
    hipEventElapsedTime(t_hog, a0, a1);
    hipEventElapsedTime(t_tiny, a0, b1);
    return *t_tiny > 0.5f * *t_hog;
  };
  float th, tt;
  conflict(nullptr, s[0], &th, &tt);      // warm-up
  printf("hog on the default stream, tiny on stream i:\n");
  for (int i = 0; i < n; ++i) {
    const bool c = conflict(nullptr, s[i], &th, &tt);
    printf("  s%-2d %s  (hog %.0f us, tiny done after %.0f us)\n", i, c ? "BLOCKED" : "free   ", th * 1e3, tt * 1e3);
  }
  printf("hog on s0, tiny on stream i:\n");
  for (int i = 1; i < n; ++i) {
    const bool c = conflict(s[0], s[i], &th, &tt);
    printf("  s%-2d %s  (hog %.0f us, tiny done after %.0f us)\n", i, c ? "BLOCKED" : "free   ", th * 1e3, tt * 1e3);
  }
  return 0;
}
