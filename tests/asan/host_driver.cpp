// AddressSanitizer driver of the HOST side of libcqlrec ("-fsanitize=address host build of the C++ op layer").
// Linked against the host-only, ASan-instrumented build of the library (replay_cql_amd/build.py::build_asan: hipcc
// --cuda-host-only -fsanitize=address -- GPU ASan is not available on the pool, and nothing here launches a kernel).
// It walks every entry point's argument validation and error reporting (CQL_REQUIRE -> cql_set_error: formatted into
// a thread-local buffer), the size / split / layout arithmetic, and the workspace-size functions over a grid of
// shapes, exactly as a caller without a GPU can reach them.  Exit code 0 and no ASan report = pass.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/cqlrec.h"

static int fails = 0;
#define EXPECT(cond)                                                         \
  do {                                                                       \
    if (!(cond)) {                                                           \
      fprintf(stderr, "host_driver: %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      ++fails;                                                               \
    }                                                                        \
  } while (0)

static void layouts() {
  const int64_t ns[] = {1, 4, 63, 64, 65, 257, 3883, 10007, 100000, 1000000};
  const int32_t ds[] = {64, 128, 256};
  for (int64_t n : ns)
    for (int32_t d : ds) {
      cqlrec_layout l;
      memset(&l, 0xAB, sizeof l);
      EXPECT(cqlrec_layout_make(n, d, &l) == CQLREC_OK);
      EXPECT(l.n_items == n && l.d == d && l.off_E_in == 0);
      EXPECT(l.off_E_out >= (n + 1) * d && l.off_E_out % CQLREC_SEG_ALIGN == 0);
      EXPECT(l.off_b_out >= l.off_E_out + n * d && l.off_W1 >= l.off_b_out + n && l.total % CQLREC_SEG_ALIGN == 0);
      EXPECT(l.total >= l.off_b2 + d);
    }
  cqlrec_layout l;
  EXPECT(cqlrec_layout_make(0, 128, &l) != CQLREC_OK);
  EXPECT(cqlrec_layout_make(100, 100, &l) != CQLREC_OK);
  EXPECT(cqlrec_layout_make(100, 128, nullptr) != CQLREC_OK);
  EXPECT(strlen(cqlrec_last_error()) > 0);
}

static void workspace_sizes() {
  const int64_t users[] = {1, 63, 256, 257, 4000, 62500, 65536};
  const int64_t items[] = {1, 31, 64, 1000, 3883, 10007, 100000, 1000000};
  const int32_t ds[] = {64, 128, 256};
  for (int64_t u : users)
    for (int64_t n : items)
      for (int32_t d : ds) {
        const int64_t a = cqlrec_topk_ws_bytes(u, n, d, 10);
        EXPECT(a > 0 && a % 256 == 0);
        EXPECT(cqlrec_topk_ws_bytes(u, n, d, 2048) >= 0);
        if (u <= 4096) {
          EXPECT(cqlrec_qhead_ws_bytes(u, n, d) > 0);
          EXPECT(cqlrec_qhead_bwd_ws_bytes(u, n, d) > 0);
          EXPECT(cqlrec_qhead_fused_ws_bytes(u, n, d) > 0);
          EXPECT(cqlrec_train_ws_bytes((int32_t)u, n, d, 50) > 0);
          EXPECT(cqlrec_gather_pool_bwd_ws_bytes(u, 50, d) > 0);
          EXPECT(cqlrec_encoder_bwd_ws_bytes(u, d) > 0);
        }
      }
  EXPECT(cqlrec_build_csr_ws_bytes(0) >= 0);
  EXPECT(cqlrec_build_csr_ws_bytes(1000003) > 0);
  EXPECT(cqlrec_eval_topk_ws_bytes(4096, 3) >= 0);
}

// every call below must be REFUSED by the argument checks in front of any launch (no GPU is needed, none is touched)
static void refused_calls() {
  std::vector<uint16_t> h16(64 * 128);
  std::vector<float> f32(4096);
  std::vector<int32_t> i32(4096);
  std::vector<int64_t> i64(4096);
  std::vector<unsigned char> ws(1 << 16);
  void* p = ws.data();
  EXPECT(cqlrec_score_topk(nullptr, 8, h16.data(), f32.data(), 64, 128, nullptr, nullptr, nullptr, nullptr, 4, p, 1 << 16,
                           i32.data(), f32.data(), i32.data(), nullptr) == CQLREC_ERR_INVALID);
  EXPECT(cqlrec_score_topk(h16.data(), 8, h16.data(), f32.data(), 64, 100, nullptr, nullptr, nullptr, nullptr, 4, p, 1 << 16,
                           i32.data(), f32.data(), i32.data(), nullptr) == CQLREC_ERR_INVALID);      // d unsupported
  EXPECT(cqlrec_score_topk(h16.data(), 8, h16.data(), f32.data(), 64, 128, nullptr, nullptr, nullptr, nullptr, 0, p, 1 << 16,
                           i32.data(), f32.data(), i32.data(), nullptr) == CQLREC_ERR_INVALID);      // k out of range
  EXPECT(cqlrec_score_topk(h16.data(), 8, h16.data(), f32.data(), 64, 128, nullptr, i64.data(), nullptr, nullptr, 4, p,
                           1 << 16, i32.data(), f32.data(), i32.data(), nullptr) == CQLREC_ERR_INVALID);   // seen_items NULL
  EXPECT(cqlrec_score_topk(h16.data(), 4000, h16.data(), f32.data(), 5000, 128, nullptr, nullptr, nullptr, nullptr, 4, p, 16,
                           i32.data(), f32.data(), i32.data(), nullptr) == CQLREC_ERR_INVALID);      // workspace too small
  EXPECT(cqlrec_score_topk_phase(h16.data(), 8, h16.data(), f32.data(), 64, 128, nullptr, nullptr, nullptr, nullptr, 4, p,
                                 1 << 16, i32.data(), f32.data(), i32.data(), 77, nullptr) == CQLREC_ERR_INVALID);
  EXPECT(strstr(cqlrec_last_error(), "phase") != nullptr);
  EXPECT(cqlrec_adam_ema(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 64, 1e-3f, 1.f, 0.9f, 0.999f, 1e-8f,
                         0.005f, 1, nullptr) == CQLREC_ERR_INVALID);
  EXPECT(cqlrec_adam_ema(f32.data(), f32.data(), f32.data(), f32.data(), f32.data(), h16.data(), h16.data(), 6, 1e-3f, 1.f,
                         0.9f, 0.999f, 1e-8f, 0.005f, 1, nullptr) == CQLREC_ERR_INVALID);             // n % 4 != 0
  EXPECT(cqlrec_qhead_fwd(nullptr, 8, h16.data(), f32.data(), 64, 128, 0, p, 1 << 16, f32.data(), nullptr, nullptr,
                          nullptr) == CQLREC_ERR_INVALID);
  EXPECT(cqlrec_gather_dot(nullptr, h16.data(), f32.data(), i32.data(), 8, 128, f32.data(), nullptr) == CQLREC_ERR_INVALID);
  EXPECT(cqlrec_linear_bf16(nullptr, h16.data(), f32.data(), 8, 128, 1, nullptr, h16.data(), nullptr) == CQLREC_ERR_INVALID);
  EXPECT(cqlrec_gather_pool_fwd(nullptr, i64.data(), i32.data(), i32.data(), nullptr, 0, 8, 50, 128, f32.data(), nullptr,
                                nullptr, nullptr) == CQLREC_ERR_INVALID);
  EXPECT(cqlrec_build_csr(nullptr, nullptr, nullptr, nullptr, 10, 4, p, 1 << 16, i64.data(), i32.data(), f32.data(),
                          nullptr) == CQLREC_ERR_INVALID);
  EXPECT(cqlrec_train_steps(nullptr, 0, 1, nullptr, nullptr) == CQLREC_ERR_INVALID);
  EXPECT(cqlrec_train_views_get(nullptr, 0, nullptr) == CQLREC_ERR_INVALID);
  EXPECT(cqlrec_prof_enable(0) == CQLREC_OK);
  EXPECT(cqlrec_prof_select(0xFFFFFFFFu) == CQLREC_OK);
  EXPECT(cqlrec_set_concurrency(1) == CQLREC_OK);
  EXPECT(cqlrec_abi_version() == CQLREC_ABI_VERSION);
  EXPECT(cqlrec_aux_stream(-1) == nullptr && cqlrec_aux_stream(CQLREC_AUX_STREAMS) == nullptr);
  EXPECT(cqlrec_aux_stream(0) == nullptr);       // before cqlrec_runtime_init (which needs a GPU)
  EXPECT(cqlrec_runtime_probe_count() == 0);
}

// the error message buffer is thread-local: concurrent failing calls must not trample each other's text
static void threads() {
  std::vector<std::thread> th;
  for (int t = 0; t < 4; ++t)
    th.emplace_back([t] {
      for (int i = 0; i < 200; ++i) {
        cqlrec_layout l;
        if (t & 1) {
          if (cqlrec_layout_make(100, 100 + t, &l) == CQLREC_OK) ++fails;
          if (strstr(cqlrec_last_error(), std::to_string(100 + t).c_str()) == nullptr) ++fails;
        } else if (cqlrec_layout_make(100 + i, 128, &l) != CQLREC_OK) {
          ++fails;
        }
      }
    });
  for (auto& x : th) x.join();
}

int main() {
  layouts();
  workspace_sizes();
  refused_calls();
  threads();
  if (fails) {
    fprintf(stderr, "host_driver: %d expectation(s) failed\n", fails);
    return 1;
  }
  printf("host_driver: ok\n");
  return 0;
}
