// torch.ops.cqlrec.* -- PyTorch-ROCm custom-op registration of the hot path ("PyTorch-ROCm custom ops in one
// .so, namespace torch.ops.cqlrec": gather_pool_fwd/bwd, qhead_lse_fwd/bwd, qhead_gather_dot, score_topk,
// fused_adam_ema).  A thin shim: every op validates its tensors (TORCH_CHECK -> Python RuntimeError), allocates its
// outputs and scratch through torch's caching allocator, and calls the C ABI of include/cqlrec.h on the CURRENT HIP
// stream.  No arithmetic lives here; the kernels are in libcqlrec.so, which this library links against.
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <tuple>

#include "../../include/cqlrec.h"

namespace {
using at::Tensor;
using c10::optional;

cqlrec_stream cur_stream(const Tensor& t) {
  return (cqlrec_stream)c10::hip::getCurrentHIPStream(t.device().index()).stream();
}
void ok(int rc, const char* what) { TORCH_CHECK(rc == CQLREC_OK, "cqlrec.", what, ": ", cqlrec_last_error()); }
// Every op opens with `OpDevice dev_(first tensor)`: the op's device becomes the CURRENT device for its duration (the
// library's per-device tables -- side streams, LDS opt-ins -- are indexed by hipGetDevice(), and a launch must go to
// the device whose stream it is given), and every further tensor argument must live on that same device.
thread_local c10::Device g_op_device(c10::DeviceType::CPU);
struct OpDevice {
  c10::DeviceGuard guard;      // (generic guard: on a ROCm build the "cuda" device type is served by the HIP implementation)
  explicit OpDevice(const Tensor& first) : guard(first.device()) {
    TORCH_CHECK(first.is_cuda(), "cqlrec ops take GPU tensors");
    g_op_device = first.device();
  }
};
void dev_contig(const Tensor& t, at::ScalarType st, const char* name) {
  TORCH_CHECK(t.is_cuda(), name, " must live on the GPU");
  TORCH_CHECK(t.device() == g_op_device, name, " is on ", t.device(), " but the op's first tensor is on ", g_op_device,
              ": all tensor arguments must share one device");
  TORCH_CHECK(t.scalar_type() == st, name, " must be ", c10::toString(st), ", got ", c10::toString(t.scalar_type()));
  TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}
const uint16_t* bf(const Tensor& t) { return reinterpret_cast<const uint16_t*>(t.data_ptr()); }
uint16_t* bfm(Tensor& t) { return reinterpret_cast<uint16_t*>(t.data_ptr()); }
template <typename T>
const T* optp(const optional<Tensor>& t) { return (t.has_value() && t->defined()) ? t->data_ptr<T>() : nullptr; }
Tensor scratch(int64_t bytes, const Tensor& like) {
  return at::empty({bytes}, like.options().dtype(at::kByte));
}

// h0 (fp32) and h0_b (bf16) of n states: state i = last min(end_i, L) items of users[i] before ends[i] + end_delta
// ID-RANGE CONTRACT (gather_pool_fwd / gather_pool_bwd): items in [0, E_in_b.size(0) - 1) resp. [0, n_items), users in
// [0, offsets.numel() - 1), ends within the user's row -- the kernels index E_in / g_E_in with them unguarded, and
// checking here would cost a device->host sync per call.  Validate a log ONCE where it enters (CQLCore.set_log and
// CQL._device_states do: one min/max reduction), not per op.
std::tuple<Tensor, Tensor> gather_pool_fwd(const Tensor& E_in_b, const Tensor& offsets, const Tensor& items,
                                           const Tensor& users, const optional<Tensor>& ends, int64_t end_delta,
                                           int64_t window) {
  OpDevice dev_(E_in_b);
  dev_contig(E_in_b, at::kBFloat16, "E_in_b");
  dev_contig(offsets, at::kLong, "offsets");
  dev_contig(items, at::kInt, "items");
  dev_contig(users, at::kInt, "users");
  if (ends.has_value() && ends->defined()) dev_contig(*ends, at::kInt, "ends");
  TORCH_CHECK(E_in_b.dim() == 2, "E_in_b must be [N+1, d]");
  const int64_t n = users.numel(), d = E_in_b.size(1);
  Tensor h0 = at::empty({n, d}, E_in_b.options().dtype(at::kFloat));
  Tensor h0b = at::empty({n, d}, E_in_b.options());
  if (n)
    ok(cqlrec_gather_pool_fwd(bf(E_in_b), offsets.data_ptr<int64_t>(), items.data_ptr<int32_t>(),
                              users.data_ptr<int32_t>(), optp<int32_t>(ends), (int32_t)end_delta, n, (int32_t)window,
                              (int32_t)d, h0.data_ptr<float>(), bfm(h0b), nullptr, cur_stream(E_in_b)),
       "gather_pool_fwd");
  return {h0, h0b};
}

// g_E_in [n_rows x d] (zero-initialised here) += dh0[i] / len_i over the window rows; deterministic sorted form
Tensor gather_pool_bwd(const Tensor& dh0, const Tensor& offsets, const Tensor& items, const Tensor& users,
                       const optional<Tensor>& ends, int64_t end_delta, int64_t window, int64_t n_items) {
  OpDevice dev_(dh0);
  dev_contig(dh0, at::kFloat, "dh0");
  dev_contig(offsets, at::kLong, "offsets");
  dev_contig(items, at::kInt, "items");
  dev_contig(users, at::kInt, "users");
  if (ends.has_value() && ends->defined()) dev_contig(*ends, at::kInt, "ends");
  TORCH_CHECK(dh0.dim() == 2 && dh0.size(0) == users.numel(), "dh0 must be [n_states, d]");
  const int64_t n = users.numel(), d = dh0.size(1);
  Tensor g = at::zeros({n_items + 1, d}, dh0.options());
  if (n) {
    const int64_t wsb = cqlrec_gather_pool_bwd_ws_bytes(n, (int32_t)window, (int32_t)d);
    Tensor ws = scratch(wsb, dh0);
    ok(cqlrec_gather_pool_bwd_sorted(dh0.data_ptr<float>(), offsets.data_ptr<int64_t>(), items.data_ptr<int32_t>(),
                                     users.data_ptr<int32_t>(), optp<int32_t>(ends), (int32_t)end_delta, n,
                                     (int32_t)window, (int32_t)d, n_items, ws.data_ptr(), wsb, g.data_ptr<float>(),
                                     cur_stream(dh0)),
       "gather_pool_bwd");
  }
  return g;
}

// (lse, -lse*log2e) of Q = H_b E_out_b^T + b_out per row; the score matrix never reaches HBM
std::tuple<Tensor, Tensor> qhead_lse_fwd(const Tensor& H_b, const Tensor& E_out_b, const Tensor& b_out) {
  OpDevice dev_(H_b);
  dev_contig(H_b, at::kBFloat16, "H_b");
  dev_contig(E_out_b, at::kBFloat16, "E_out_b");
  dev_contig(b_out, at::kFloat, "b_out");
  TORCH_CHECK(H_b.dim() == 2 && E_out_b.dim() == 2 && H_b.size(1) == E_out_b.size(1) && b_out.numel() == E_out_b.size(0),
              "shapes: H_b [rows, d], E_out_b [N, d], b_out [N]");
  const int64_t rows = H_b.size(0), n = E_out_b.size(0), d = H_b.size(1);
  Tensor lse = at::empty({rows}, b_out.options()), nlse2 = at::empty({rows}, b_out.options());
  const int64_t wsb = cqlrec_qhead_ws_bytes(rows, n, (int32_t)d);
  Tensor ws = scratch(wsb, H_b);
  ok(cqlrec_qhead_fwd(bf(H_b), rows, bf(E_out_b), b_out.data_ptr<float>(), n, (int32_t)d, CQLREC_QHEAD_LSE,
                      ws.data_ptr(), wsb, lse.data_ptr<float>(), nullptr, nlse2.data_ptr<float>(), cur_stream(H_b)),
     "qhead_lse_fwd");
  return {lse, nlse2};
}

// (max_j Q, argmax_j Q (ties -> smallest j)) per row
std::tuple<Tensor, Tensor> qhead_argmax_fwd(const Tensor& H_b, const Tensor& E_out_b, const Tensor& b_out) {
  OpDevice dev_(H_b);
  dev_contig(H_b, at::kBFloat16, "H_b");
  dev_contig(E_out_b, at::kBFloat16, "E_out_b");
  dev_contig(b_out, at::kFloat, "b_out");
  TORCH_CHECK(H_b.dim() == 2 && E_out_b.dim() == 2 && H_b.size(1) == E_out_b.size(1) && b_out.numel() == E_out_b.size(0),
              "shapes: H_b [rows, d], E_out_b [N, d], b_out [N]");
  const int64_t rows = H_b.size(0), n = E_out_b.size(0), d = H_b.size(1);
  Tensor vmax = at::empty({rows}, b_out.options()), imax = at::empty({rows}, b_out.options().dtype(at::kInt));
  const int64_t wsb = cqlrec_qhead_ws_bytes(rows, n, (int32_t)d);
  Tensor ws = scratch(wsb, H_b);
  ok(cqlrec_qhead_fwd(bf(H_b), rows, bf(E_out_b), b_out.data_ptr<float>(), n, (int32_t)d, CQLREC_QHEAD_ARGMAX,
                      ws.data_ptr(), wsb, vmax.data_ptr<float>(), imax.data_ptr<int32_t>(), nullptr, cur_stream(H_b)),
     "qhead_argmax_fwd");
  return {vmax, imax};
}

// (dH, g_E_out, g_b_out) of dQ = scale * bf16(softmax) + coef * onehot(act)
std::tuple<Tensor, Tensor, Tensor> qhead_lse_bwd(const Tensor& H_b, const Tensor& nlse2, const Tensor& coef,
                                                 const Tensor& act, const Tensor& E_out_b, const Tensor& b_out,
                                                 double scale) {
  OpDevice dev_(H_b);
  dev_contig(H_b, at::kBFloat16, "H_b");
  dev_contig(nlse2, at::kFloat, "nlse2");
  dev_contig(coef, at::kFloat, "coef");
  dev_contig(act, at::kInt, "act");
  dev_contig(E_out_b, at::kBFloat16, "E_out_b");
  dev_contig(b_out, at::kFloat, "b_out");
  const int64_t rows = H_b.size(0), n = E_out_b.size(0), d = H_b.size(1);
  TORCH_CHECK(nlse2.numel() == rows && coef.numel() == rows && act.numel() == rows, "nlse2 / coef / act must be [rows]");
  Tensor dH = at::empty({rows, d}, b_out.options()), gE = at::empty({n, d}, b_out.options()),
         gb = at::empty({n}, b_out.options());
  const int64_t wsb = cqlrec_qhead_bwd_ws_bytes(rows, n, (int32_t)d);
  Tensor ws = scratch(wsb, H_b);
  ok(cqlrec_qhead_bwd(bf(H_b), nlse2.data_ptr<float>(), coef.data_ptr<float>(), act.data_ptr<int32_t>(), rows,
                      bf(E_out_b), b_out.data_ptr<float>(), n, (int32_t)d, (float)scale, ws.data_ptr(), wsb,
                      dH.data_ptr<float>(), gE.data_ptr<float>(), gb.data_ptr<float>(), cur_stream(H_b)),
     "qhead_lse_bwd");
  return {dH, gE, gb};
}

// out[r] = <H_b[r], E_b[idx[r]]> + b[idx[r]]
Tensor qhead_gather_dot(const Tensor& H_b, const Tensor& E_b, const Tensor& b, const Tensor& idx) {
  OpDevice dev_(H_b);
  dev_contig(H_b, at::kBFloat16, "H_b");
  dev_contig(E_b, at::kBFloat16, "E_b");
  dev_contig(b, at::kFloat, "b");
  dev_contig(idx, at::kInt, "idx");
  TORCH_CHECK(H_b.dim() == 2 && idx.numel() == H_b.size(0) && E_b.size(1) == H_b.size(1), "shapes");
  Tensor out = at::empty({H_b.size(0)}, b.options());
  if (H_b.size(0))
    ok(cqlrec_gather_dot(bf(H_b), bf(E_b), b.data_ptr<float>(), idx.data_ptr<int32_t>(), H_b.size(0),
                         (int32_t)H_b.size(1), out.data_ptr<float>(), cur_stream(H_b)),
       "qhead_gather_dot");
  return out;
}

// (idx [n x k], val [n x k], cnt [n]): the k best (score desc, item id asc) admissible items per state vector
std::tuple<Tensor, Tensor, Tensor> score_topk(const Tensor& H_b, const Tensor& E_b, const Tensor& b, int64_t k,
                                              const optional<Tensor>& item_ids, const optional<Tensor>& seen_off,
                                              const optional<Tensor>& seen_items, const optional<Tensor>& seen_rows) {
  OpDevice dev_(H_b);
  dev_contig(H_b, at::kBFloat16, "H_b");
  dev_contig(E_b, at::kBFloat16, "E_b");
  dev_contig(b, at::kFloat, "b");
  if (item_ids.has_value() && item_ids->defined()) dev_contig(*item_ids, at::kInt, "item_ids");
  if (seen_off.has_value() && seen_off->defined()) dev_contig(*seen_off, at::kLong, "seen_off");
  if (seen_items.has_value() && seen_items->defined()) dev_contig(*seen_items, at::kInt, "seen_items");
  if (seen_rows.has_value() && seen_rows->defined()) dev_contig(*seen_rows, at::kInt, "seen_rows");
  TORCH_CHECK(H_b.dim() == 2 && E_b.dim() == 2 && H_b.size(1) == E_b.size(1) && b.numel() == E_b.size(0), "shapes");
  TORCH_CHECK(k > 0, "k must be positive");
  const int64_t n = H_b.size(0), nc = E_b.size(0), d = H_b.size(1);
  Tensor idx = at::empty({n, k}, b.options().dtype(at::kInt)), val = at::empty({n, k}, b.options()),
         cnt = at::empty({n}, b.options().dtype(at::kInt));
  if (n) {
    const int64_t wsb = cqlrec_topk_ws_bytes(n, nc, (int32_t)d, (int32_t)k);
    Tensor ws = scratch(wsb, H_b);
    ok(cqlrec_score_topk(bf(H_b), n, bf(E_b), b.data_ptr<float>(), nc, (int32_t)d, optp<int32_t>(item_ids),
                         optp<int64_t>(seen_off), optp<int32_t>(seen_items), optp<int32_t>(seen_rows), (int32_t)k,
                         ws.data_ptr(), wsb, idx.data_ptr<int32_t>(), val.data_ptr<float>(), cnt.data_ptr<int32_t>(),
                         cur_stream(H_b)),
       "score_topk");
  }
  return {idx, val, cnt};
}

// in place: Adam + Polyak target + both bf16 shadows (+ gradient zeroing) over flat buffers
void fused_adam_ema(Tensor theta, Tensor grads, Tensor m, Tensor v, Tensor target, Tensor theta_b, Tensor target_b,
                    double step_size, double sqrt_bc2, double beta1, double beta2, double eps, double tau,
                    bool zero_grads) {
  OpDevice dev_(theta);
  for (const Tensor* t : {&theta, &grads, &m, &v, &target}) dev_contig(*t, at::kFloat, "theta/grads/m/v/target");
  dev_contig(theta_b, at::kBFloat16, "theta_b");
  dev_contig(target_b, at::kBFloat16, "target_b");
  const int64_t n = theta.numel();
  TORCH_CHECK(grads.numel() == n && m.numel() == n && v.numel() == n && target.numel() == n && theta_b.numel() == n &&
                  target_b.numel() == n, "all buffers must have the same number of elements");
  if (n)
    ok(cqlrec_adam_ema(theta.data_ptr<float>(), grads.data_ptr<float>(), m.data_ptr<float>(), v.data_ptr<float>(),
                       target.data_ptr<float>(), bfm(theta_b), bfm(target_b), n, (float)step_size, (float)sqrt_bc2,
                       (float)beta1, (float)beta2, (float)eps, (float)tau, zero_grads ? 1 : 0, cur_stream(theta)),
       "fused_adam_ema");
}
}  // namespace

TORCH_LIBRARY(cqlrec, m) {
  m.def("gather_pool_fwd(Tensor E_in_b, Tensor offsets, Tensor items, Tensor users, Tensor? ends, int end_delta, "
        "int window) -> (Tensor, Tensor)");
  m.def("gather_pool_bwd(Tensor dh0, Tensor offsets, Tensor items, Tensor users, Tensor? ends, int end_delta, "
        "int window, int n_items) -> Tensor");
  m.def("qhead_lse_fwd(Tensor H_b, Tensor E_out_b, Tensor b_out) -> (Tensor, Tensor)");
  m.def("qhead_argmax_fwd(Tensor H_b, Tensor E_out_b, Tensor b_out) -> (Tensor, Tensor)");
  m.def("qhead_lse_bwd(Tensor H_b, Tensor nlse2, Tensor coef, Tensor act, Tensor E_out_b, Tensor b_out, float scale) "
        "-> (Tensor, Tensor, Tensor)");
  m.def("qhead_gather_dot(Tensor H_b, Tensor E_b, Tensor b, Tensor idx) -> Tensor");
  m.def("score_topk(Tensor H_b, Tensor E_b, Tensor b, int k, Tensor? item_ids=None, Tensor? seen_off=None, "
        "Tensor? seen_items=None, Tensor? seen_rows=None) -> (Tensor, Tensor, Tensor)");
  m.def("fused_adam_ema(Tensor(a!) theta, Tensor(b!) grads, Tensor(c!) m, Tensor(d!) v, Tensor(e!) target, "
        "Tensor(f!) theta_b, Tensor(g!) target_b, float step_size, float sqrt_bc2, float beta1, float beta2, float eps, "
        "float tau, bool zero_grads) -> ()");
}

// ROCm builds of PyTorch dispatch HIP tensors under the CUDA key
TORCH_LIBRARY_IMPL(cqlrec, CUDA, m) {
  m.impl("gather_pool_fwd", &gather_pool_fwd);
  m.impl("gather_pool_bwd", &gather_pool_bwd);
  m.impl("qhead_lse_fwd", &qhead_lse_fwd);
  m.impl("qhead_argmax_fwd", &qhead_argmax_fwd);
  m.impl("qhead_lse_bwd", &qhead_lse_bwd);
  m.impl("qhead_gather_dot", &qhead_gather_dot);
  m.impl("score_topk", &score_topk);
  m.impl("fused_adam_ema", &fused_adam_ema);
}
