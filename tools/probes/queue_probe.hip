// Which HIP streams of a process actually run concurrently?  (r3: the training step lost its concurrency depending on
// WHEN its internal streams were created relative to other work of the process.)
//   queue_probe MODE   MODE: early = create the test streams first, then 200 ms of null-stream work, then test
//                            late  = null-stream work first, then create the streams, then test
//                            early_touch = as early, but one marker per stream right after creation
// Test: a ~200 us spin kernel on each of {null, s0, s1, s2}; wall time of launching all four at once vs one alone.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <chrono>

__global__ void spin(long long cycles, int* sink) {
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (sink && threadIdx.x == 999) *sink = 1;
}
__global__ void fill(float* p, size_t n, float v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "late";
  hipStream_t s[3];
  float* buf;
  const size_t n = 256u << 20;
  hipMalloc(&buf, n * 4);
  auto make = [&]() {
    for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    if (!strcmp(mode, "early_touch")) {
      hipEvent_t e;
      hipEventCreateWithFlags(&e, hipEventDisableTiming);
      for (auto& x : s) hipEventRecord(e, x);
      hipDeviceSynchronize();
    }
  };
  auto work = [&]() {
    for (int i = 0; i < 100; ++i) fill<<<2048, 256>>>(buf, n, (float)i);
    hipDeviceSynchronize();
  };
  if (!strncmp(mode, "early", 5)) { make(); work(); } else { work(); make(); }
  const long long cyc = 2000000;   // shader clock: ~1 ms
  auto run = [&](int k) {        // k streams at once (null first)
    hipDeviceSynchronize();
    const double t0 = now();
    spin<<<1, 64>>>(cyc, nullptr);
    for (int i = 0; i < k - 1; ++i) spin<<<1, 64, 0, s[i]>>>(cyc, nullptr);
    hipDeviceSynchronize();
    return (now() - t0) * 1e6;
  };
  run(4);
  const double one = run(1), four = run(4), two = run(2);
  // pairs among the created streams
  auto pair = [&](int a, int b) {
    hipDeviceSynchronize();
    const double t0 = now();
    spin<<<1, 64, 0, s[a]>>>(cyc, nullptr);
    spin<<<1, 64, 0, s[b]>>>(cyc, nullptr);
    hipDeviceSynchronize();
    return (now() - t0) * 1e6;
  };
  printf("%s: one=%.0f us  null+s0=%.0f  all four=%.0f  s0+s1=%.0f s0+s2=%.0f s1+s2=%.0f\n", mode, one, two, four, pair(0, 1),
         pair(0, 2), pair(1, 2));
  return 0;
}
