"""CQLCore: arrays-in / arrays-out host driver of the HIP hot path (no Spark, no pandas).

torch is plumbing only here: it owns device memory, the HIP stream and (for data parallelism) the RCCL process
group.  Every arithmetic step of fit and predict is a libcqlrec.so kernel reached through the C ABI
(include/cqlrec.h); there is no CPU or torch fallback -- a missing library raises (see _native.load).

Role in the reference's structure: this is what the body of a `CQL._fit` / `CQL._predict` calls where
NeuroMF calls `TorchRecommender.train` (replay/models/base_torch_rec.py:57-98) and the per-user pandas UDF
(replay/models/base_torch_rec.py:120-149)."""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass, asdict
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _native as N


@dataclass
class CQLHyper:
    d: int = 128
    window: int = 50
    batch: int = 4096
    gamma: float = 0.99
    alpha: float = 1.0
    lr: float = 1e-3
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    tau: float = 0.005
    seed: int = 0


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class CQLCore:
    """Device-resident CQL model + trainer + scorer for one GPU (one process per GPU under data parallelism)."""

    def __init__(self, n_items: int, hyper: Optional[CQLHyper] = None, device: Optional[torch.device] = None,
                 rank: int = 0, world: int = 1, process_group=None, init_seed: int = 7,
                 shard_optimizer: bool = False):
        self.lib = N.load()
        if not torch.cuda.is_available():
            raise N.CqlrecError("CQLCore needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path")
        self.hyper = hyper or CQLHyper()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        with torch.cuda.device(self.device):
            N.runtime_init()          # the library's streams first: see cqlrec_runtime_init in include/cqlrec.h
        self.rank, self.world, self.pg = int(rank), int(world), process_group
        self.n_items = int(n_items)
        self.layout = N.make_layout(self.n_items, self.hyper.d)
        P = int(self.layout.total)
        dev = self.device
        self.theta = torch.zeros(P, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(P, dtype=torch.float32, device=dev)
        self.adam_m = torch.zeros(P, dtype=torch.float32, device=dev)
        self.adam_v = torch.zeros(P, dtype=torch.float32, device=dev)
        self.target = torch.zeros(P, dtype=torch.float32, device=dev)
        self.theta_b = torch.zeros(P, dtype=torch.bfloat16, device=dev)
        self.target_b = torch.zeros(P, dtype=torch.bfloat16, device=dev)
        self.step = 0
        self._csr: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None
        self._ws: Optional[torch.Tensor] = None
        self._ctx: Optional[N.TrainCtx] = None
        self._side: Optional[torch.cuda.Stream] = None
        # Row-sharded optimizer (the cfg5 variant): rank r keeps the fp32 masters, Adam moments and target
        # of rows [r*R, (r+1)*R) of E_in and E_out up to date (R = N // W; the few remaining rows, b_out and the encoder
        # stay replicated), gradients are reduce-scattered, the bf16 shadows all-gathered.  Same bytes on the links as
        # the all-reduce, 1/W of the Adam traffic per GPU.  Opt-in; needs world > 1 and N >= W.
        self.shard_optimizer = bool(shard_optimizer) and self.world > 1 and self.n_items >= self.world
        self._gshard: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
        self._tailpack: Optional[torch.Tensor] = None
        self._agstage: Optional[Tuple[torch.Tensor, ...]] = None
        # Replicated variant, how the gradient SUM travels: "allreduce" (one RCCL all-reduce per half) or "rsag" -- an
        # explicit reduce-scatter into a persistent 1/W shard + all-gather back (each GPU exchanges S/W
        # with each of its 7 xGMI peers per phase).  Same bytes, same sums in a different association order across
        # ranks (every rank still ends with identical bits: each element is reduced on ONE rank).  Unmeasured on
        # hardware (no multi-GPU box in this build); opt-in via CQL_DP_EXCHANGE=rsag.
        import os as _os
        self._exchange = _os.environ.get("CQL_DP_EXCHANGE", "allreduce")
        if self._exchange not in ("allreduce", "rsag"):
            raise ValueError("CQL_DP_EXCHANGE must be allreduce or rsag")
        self._rsag_shards: Dict[Tuple[int, int], torch.Tensor] = {}
        self._topk_side: Optional[torch.cuda.Stream] = None
        self.init_params(init_seed)

    # ------------------------------------------------------------------ parameters
    def segment(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        shp = self.layout.shape(name)
        off = self.layout.offset(name)
        return flat[off: off + int(np.prod(shp))].view(*shp)

    def init_params(self, seed: int = 7) -> None:
        """E_in, E_out ~ N(0, 1/d); W ~ xavier_normal; biases 0; PAD row 0."""
        d, n = self.hyper.d, self.n_items
        g = torch.Generator(device="cpu").manual_seed(int(seed))
        flat = torch.zeros(int(self.layout.total), dtype=torch.float32)
        for name, std in (("E_in", 1.0 / math.sqrt(d)), ("E_out", 1.0 / math.sqrt(d)),
                          ("W1", math.sqrt(2.0 / (2 * d))), ("W2", math.sqrt(2.0 / (2 * d)))):
            shp = self.layout.shape(name)
            off = self.layout.offset(name)
            rows = n if name == "E_in" else shp[0]
            flat[off: off + rows * shp[1]] = torch.randn(rows * shp[1], generator=g) * std
        self.load_flat(flat)

    def load_flat(self, theta, target=None, adam_m=None, adam_v=None, step: int = 0) -> None:
        """Install fp32 masters (numpy or torch, length layout.total) and refresh the bf16 shadows on device."""
        def put(dst, src):
            src = torch.as_tensor(np.asarray(src) if not torch.is_tensor(src) else src, dtype=torch.float32)
            if src.numel() != dst.numel():
                raise ValueError(f"flat buffer has {src.numel()} elements, layout needs {dst.numel()}")
            dst.copy_(src.to(self.device))
        put(self.theta, theta)
        put(self.target, theta if target is None else target)
        self.adam_m.zero_() if adam_m is None else put(self.adam_m, adam_m)
        self.adam_v.zero_() if adam_v is None else put(self.adam_v, adam_v)
        self.grads.zero_()
        self.step = int(step)
        self.refresh_shadows()

    def refresh_shadows(self) -> None:
        P = int(self.layout.total)
        N.check(self.lib.cqlrec_cast_bf16(_ptr(self.theta), _ptr(self.theta_b), P, _stream()), "cast_bf16")
        N.check(self.lib.cqlrec_cast_bf16(_ptr(self.target), _ptr(self.target_b), P, _stream()), "cast_bf16")

    def state_dict(self) -> Dict[str, object]:
        """Full fp32 state.  With the row-sharded optimizer this is a collective: every rank must call it."""
        self.sync_full_state()
        return {"theta": self.theta.cpu(), "target": self.target.cpu(), "adam_m": self.adam_m.cpu(),
                "adam_v": self.adam_v.cpu(), "step": self.step, "n_items": self.n_items,
                "hyper": asdict(self.hyper)}

    def load_state_dict(self, sd: Dict[str, object]) -> None:
        self.load_flat(sd["theta"], sd["target"], sd["adam_m"], sd["adam_v"], int(sd["step"]))

    # ------------------------------------------------------------------ training
    def set_log(self, offsets, items, rewards) -> None:
        """CSR by user (S2): offsets int64[U+1], items int32[nnz], rewards float32[nnz] (numpy or torch)."""
        def dev(x, dt):
            t = torch.as_tensor(np.ascontiguousarray(x) if not torch.is_tensor(x) else x)
            return t.to(device=self.device, dtype=dt).contiguous()
        offsets, items, rewards = dev(offsets, torch.int64), dev(items, torch.int32), dev(rewards, torch.float32)
        if offsets.numel() < 2 or int(offsets[-1]) != items.numel() or items.numel() != rewards.numel():
            raise ValueError("inconsistent CSR: offsets[-1] must equal len(items) == len(rewards) and U >= 1")
        if items.numel() == 0:
            raise ValueError("empty log")
        # the window gather reads E_in + item*d and both scatters write g_E_* + item*d unguarded: check the ids once
        lo, hi = torch.stack([items.min(), items.max()]).cpu().tolist()
        if lo < 0 or hi >= self.n_items:
            raise ValueError(f"item ids must lie in [0, {self.n_items}); the log holds [{int(lo)}, {int(hi)}]")
        self._csr = (offsets, items, rewards)
        self._ctx = None

    def _train_ctx(self) -> N.TrainCtx:
        if self._ctx is not None:
            return self._ctx
        if self._csr is None:
            raise RuntimeError("set_log() must be called before training")
        h = self.hyper
        need = int(self.lib.cqlrec_train_ws_bytes(h.batch, self.n_items, h.d, h.window))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        offsets, items, rewards = self._csr
        c = N.TrainCtx()
        c.layout = self.layout
        c.offsets, c.items, c.rewards = _ptr(offsets), _ptr(items), _ptr(rewards)
        c.n_users = offsets.numel() - 1
        c.theta, c.grads, c.adam_m, c.adam_v = _ptr(self.theta), _ptr(self.grads), _ptr(self.adam_m), _ptr(self.adam_v)
        c.target, c.theta_b, c.target_b = _ptr(self.target), _ptr(self.theta_b), _ptr(self.target_b)
        c.batch, c.window, c.world, c.rank = h.batch, h.window, self.world, self.rank
        c.gamma, c.alpha, c.lr, c.beta1, c.beta2, c.eps, c.tau = h.gamma, h.alpha, h.lr, h.beta1, h.beta2, h.eps, h.tau
        c.seed = h.seed
        c.ws, c.ws_bytes = _ptr(self._ws), self._ws.numel()
        self._ctx = c
        return c

    def forward_backward(self, loss_out: Optional[torch.Tensor] = None) -> None:
        """Sample + forward + loss + backward of global step `self.step` into self.grads (async)."""
        c = self._train_ctx()
        N.check(self.lib.cqlrec_train_step_fwd_bwd(C.byref(c), self.step, _ptr(loss_out), _stream()), "train_step_fwd_bwd")

    def apply_update(self) -> None:
        c = self._train_ctx()
        N.check(self.lib.cqlrec_train_step_update(C.byref(c), self.step, _stream()), "train_step_update")
        self.step += 1

    def allreduce_grads(self) -> None:
        if self.world > 1:
            from .dist import allreduce_sum_
            allreduce_sum_(self.grads, self.pg)

    def train_step(self, loss_out: Optional[torch.Tensor] = None, phased: Optional[bool] = None) -> None:
        """One full CQL step (see train_steps; joined before returning)."""
        self.train_steps(1, loss_out, phased)

    def train_steps(self, n_steps: int, losses: Optional[torch.Tensor] = None, phased: Optional[bool] = None) -> None:
        """n_steps whole steps, pipelined across steps, joined on the current stream before returning.
        losses: optional contiguous float32 device tensor with >= n_steps elements (this rank's share of the loss).

        Single rank: ONE library call, the step loop runs in C++ (cqlrec_train_steps).
        Data parallel (or phased=True): each step runs in phases around two asynchronous all-reduces --
          main stream: forward(t) . state-side backward . all-reduce(E_in, encoder grads) . Adam on that half .
                       [sampling, gathers, encoders of step t+1] . wait . Q-head kernels of step t+1 ...
          side stream: item-side backward . all-reduce(E_out, b_out grads: half of the bytes) . Adam on that half
        so the second all-reduce and its Adam run under the first half's Adam and the next step's prologue; only the
        catalogue-wide kernels of step t+1 wait for them (cqlrec_train_step_forward_after)."""
        n_steps = int(n_steps)
        if losses is not None and (losses.numel() < n_steps or losses.dtype != torch.float32 or not losses.is_contiguous()):
            raise ValueError("losses must be a contiguous float32 tensor with at least n_steps elements")
        phased = (self.world > 1) if phased is None else phased
        c = self._train_ctx()
        if self.shard_optimizer:
            self._train_steps_sharded(n_steps, losses)
            return
        if not phased:
            if self.world != 1:
                for i in range(n_steps):
                    self.forward_backward(None if losses is None else losses[i:i + 1])
                    self.allreduce_grads()
                    self.apply_update()
                return
            N.check(self.lib.cqlrec_train_steps(C.byref(c), self.step, n_steps, _ptr(losses), _stream()), "train_steps")
            self.step += n_steps
            return
        s, lay = _stream(), self.layout
        lo_a, hi_a, total = int(lay.off_E_out), int(lay.off_W1), int(lay.total)
        main = torch.cuda.current_stream()
        if self._side is None:
            with torch.cuda.device(self.device):
                self._side = N.aux_stream(0)          # library-owned (never a torch.cuda.Stream(): see _native.aux_stream)
            self._ev = [torch.cuda.Event() for _ in range(3)]       # forward done, state-side backward done, items ready
        side = self._side
        ev_fwd, ev_rest, ev_items = self._ev
        pending = False
        self._early_de = os.environ.get("CQL_EARLY_DE", "1") != "0"
        for i in range(n_steps):
            lo = None if losses is None else losses[i:i + 1]
            # the long dE_out kernel of this step is started by the forward itself, on the side stream, as soon as the
            # fused forward has produced the logsumexp -- under the arg-max pass and the loss; backward_items below adds
            # the parts that need the loss (CQL_EARLY_DE=0: all of it behind the loss)
            N.check(self.lib.cqlrec_train_step_forward_early_items(
                C.byref(c), self.step, _ptr(lo), s, ev_items.cuda_event if pending else None,
                side.cuda_stream if self._early_de else None), "train_step_forward_early_items")
            ev_fwd.record(main)
            N.check(self.lib.cqlrec_train_step_backward_rest(C.byref(c), self.step, s), "train_step_backward_rest")
            ev_rest.record(main)
            # issue order matters: the collectives of one process group execute in the order they were issued (one
            # internal RCCL stream), so the state-side all-reduce -- whose gradients are ready first -- goes first and
            # runs under the item-side kernel; the item-side all-reduce queues behind it
            with torch.cuda.stream(side):      # enqueue the long item-side kernel before the (slow to issue) collectives
                side.wait_event(ev_fwd)
                N.check(self.lib.cqlrec_train_step_backward_items(C.byref(c), self.step, side.cuda_stream),
                        "train_step_backward_items")
            work_b = [self._allreduce_async(self.grads[0:lo_a]), self._allreduce_async(self.grads[hi_a:total])]
            with torch.cuda.stream(side):
                work_a = self._allreduce_async(self.grads[lo_a:hi_a])
            for w in work_b:
                if w is not None:
                    w.wait()           # stream-side wait (RCCL): no host block
            N.check(self.lib.cqlrec_train_step_update_range(C.byref(c), self.step, hi_a, total, s), "update_range")
            N.check(self.lib.cqlrec_train_step_update_range(C.byref(c), self.step, 0, lo_a, s), "update_range")
            with torch.cuda.stream(side):
                if work_a is not None:
                    work_a.wait()
                side.wait_event(ev_rest)   # the state-side backward reads rows of the E_out shadow (coef * E_out[a])
                N.check(self.lib.cqlrec_train_step_update_range(C.byref(c), self.step, lo_a, hi_a, side.cuda_stream),
                        "update_range")
                ev_items.record(side)
            pending = True
            self.step += 1
        if pending:
            main.wait_event(ev_items)

    # ------------------------------------------------------------------ row-sharded optimizer (opt-in)
    def _shard_plan(self):
        from .dist import shard_plan
        lay = self.layout
        return shard_plan(int(lay.off_E_in), int(lay.off_E_out), int(lay.off_W1), int(lay.total), self.n_items,
                          self.hyper.d, self.world, self.rank)

    def _adam_shard(self, lo: int, hi: int, g: torch.Tensor, stream: int) -> None:
        """Adam + Polyak + shadows on elements [lo, hi) of the flat buffers with the gradient taken from `g` (the
        reduce-scattered shard); scalars exactly as cqlrec_train_step_update_range computes them."""
        c = self._train_ctx()
        t = float(self.step + 1)
        bc1 = 1.0 - float(c.beta1) ** t
        bc2 = 1.0 - float(c.beta2) ** t
        step_size = float(np.float32(float(c.lr) / bc1))
        sqrt_bc2 = float(np.float32(math.sqrt(bc2)))

        def at(tns):
            return tns.data_ptr() + tns.element_size() * lo
        N.check(self.lib.cqlrec_adam_ema(at(self.theta), g.data_ptr(), at(self.adam_m), at(self.adam_v), at(self.target),
                                         at(self.theta_b), at(self.target_b), hi - lo, step_size, sqrt_bc2, c.beta1, c.beta2,
                                         c.eps, c.tau, 0, stream), "adam_ema (shard)")

    def _train_steps_sharded(self, n_steps: int, losses: Optional[torch.Tensor]) -> None:
        from .dist import all_gather_into, reduce_scatter_sum
        import torch.distributed as dist
        c, s, P = self._train_ctx(), _stream(), self._shard_plan()
        main = torch.cuda.current_stream()
        if self._side is None:
            with torch.cuda.device(self.device):
                self._side = N.aux_stream(0)
            self._ev = [torch.cuda.Event() for _ in range(3)]
        if self._gshard is None:
            self._gshard = (torch.empty(P["n"], dtype=torch.float32, device=self.device),
                            torch.empty(P["n"], dtype=torch.float32, device=self.device))
        g_in, g_out = self._gshard
        if self._tailpack is None:
            self._tailpack = torch.empty((P["tail_in"][1] - P["tail_in"][0]) + (P["tail_enc"][1] - P["tail_enc"][0]),
                                         dtype=torch.float32, device=self.device)
        pk = self._tailpack
        if self._agstage is None:
            # persistent send buffers of the four shadow all-gathers (theta_b / target_b x E_in / E_out rows): the step
            # loop allocates nothing (a per-step clone would be held back by the collective's stream and churn the
            # caching allocator: 2 x 2 x 64 MB per step at cfg5)
            self._agstage = tuple(torch.empty(P["n"], dtype=torch.bfloat16, device=self.device) for _ in range(4))
        st_in, st_out = self._agstage[:2], self._agstage[2:]
        g_tail_in = self.grads[P["tail_in"][0]: P["tail_in"][1]]
        g_tail_enc = self.grads[P["tail_enc"][0]: P["tail_enc"][1]]
        side = self._side
        ev_fwd, ev_rest, ev_items = self._ev
        pg = self.pg

        def ar(lo, hi):
            return dist.all_reduce(self.grads[lo:hi], op=dist.ReduceOp.SUM, group=pg, async_op=True) if hi > lo else None

        def wait(w):
            if w is not None:
                w.wait()

        def upd(lo, hi, stream):
            if hi > lo:
                N.check(self.lib.cqlrec_train_step_update_range(C.byref(c), self.step, lo, hi, stream), "update_range")

        pending = False
        early_de = os.environ.get("CQL_EARLY_DE", "1") != "0"
        for i in range(n_steps):
            lo_ = None if losses is None else losses[i:i + 1]
            # (as in train_steps: the long dE_out kernel starts under the arg-max pass and the loss)
            N.check(self.lib.cqlrec_train_step_forward_early_items(
                C.byref(c), self.step, _ptr(lo_), s, ev_items.cuda_event if pending else None,
                side.cuda_stream if early_de else None), "train_step_forward_early_items")
            ev_fwd.record(main)
            N.check(self.lib.cqlrec_train_step_backward_rest(C.byref(c), self.step, s), "train_step_backward_rest")
            ev_rest.record(main)
            with torch.cuda.stream(side):      # the long item-side kernel is enqueued first: nothing below delays it
                side.wait_event(ev_fwd)
                N.check(self.lib.cqlrec_train_step_backward_items(C.byref(c), self.step, side.cuda_stream),
                        "train_step_backward_items")
            # Collectives of one group execute in the order they were issued (one internal RCCL stream).  State side first:
            # its gradients are ready first, and its whole exchange (reduce-scatter, Adam on the own rows, all-gather
            # of the shadows the next prologue reads) then runs under the item-side kernel; the item side queues behind.
            w_in = reduce_scatter_sum(g_in, self.grads[P["in_region"][0]: P["in_region"][1]], pg, async_op=True)
            # the two replicated remainders of the state side (last rows of E_in, encoder) travel as ONE all-reduce
            n1 = P["tail_in"][1] - P["tail_in"][0]
            torch.cat([g_tail_in, g_tail_enc], out=pk)                  # packed by ONE kernel
            w_t = dist.all_reduce(pk, op=dist.ReduceOp.SUM, group=pg, async_op=True)
            wait(w_in), wait(w_t)
            self._adam_shard(P["in_own"][0], P["in_own"][1], g_in, s)
            torch._foreach_copy_([g_tail_in, g_tail_enc], [pk[:n1], pk[n1:]])     # ... and unpacked by one
            upd(P["tail_in"][0], P["tail_in"][1], s)
            upd(P["tail_enc"][0], P["tail_enc"][1], s)
            self.grads[P["in_region"][0]: P["in_region"][1]].zero_()
            w_ag = []
            for buf, stg in zip((self.theta_b, self.target_b), st_in):
                stg.copy_(buf[P["in_own"][0]: P["in_own"][1]])
                w_ag.append(all_gather_into(buf[P["in_region"][0]: P["in_region"][1]], stg, pg, async_op=True))
            with torch.cuda.stream(side):
                w_out = reduce_scatter_sum(g_out, self.grads[P["out_region"][0]: P["out_region"][1]], pg, async_op=True)
                w_t3 = ar(*P["tail_out"])
                wait(w_out), wait(w_t3)
                side.wait_event(ev_rest)   # the state-side backward reads rows of the E_out shadow
                self._adam_shard(P["out_own"][0], P["out_own"][1], g_out, side.cuda_stream)
                upd(P["tail_out"][0], P["tail_out"][1], side.cuda_stream)
                self.grads[P["out_region"][0]: P["out_region"][1]].zero_()
                for buf, stg in zip((self.theta_b, self.target_b), st_out):
                    stg.copy_(buf[P["out_own"][0]: P["out_own"][1]])
                    wait(all_gather_into(buf[P["out_region"][0]: P["out_region"][1]], stg, pg, async_op=True))
                ev_items.record(side)
            for w in w_ag:      # the next prologue (this stream) reads the gathered E_in shadows
                wait(w)
            pending = True
            self.step += 1
        if pending:
            main.wait_event(ev_items)

    def sync_full_state(self) -> None:
        """Row-sharded optimizer: bring the fp32 masters, Adam moments and target of every rank's rows to all ranks
        (checkpointing / inspection).  No-op otherwise."""
        if not self.shard_optimizer:
            return
        from .dist import all_gather_into
        P = self._shard_plan()
        for buf in (self.theta, self.adam_m, self.adam_v, self.target):
            for reg, own in (("in_region", "in_own"), ("out_region", "out_own")):
                all_gather_into(buf[P[reg][0]: P[reg][1]], buf[P[own][0]: P[own][1]].clone(), self.pg)   # (not per step)

    def _allreduce_async(self, t: torch.Tensor):
        """SUM over ranks of a contiguous piece of the flat gradient buffer, in place, asynchronous (returns the work to
        wait on, or None)."""
        if self.world <= 1:
            return None
        import torch.distributed as dist
        if self._exchange == "rsag" and t.numel() % self.world == 0 and t.numel() > 0:
            from .dist import all_gather_into, reduce_scatter_sum
            key = (t.data_ptr(), t.numel())
            shard = self._rsag_shards.get(key)
            if shard is None:           # persistent: one per region, created on first use
                shard = self._rsag_shards[key] = torch.empty(t.numel() // self.world, dtype=t.dtype, device=t.device)
            reduce_scatter_sum(shard, t, self.pg, async_op=True)       # same group: executes in issue order
            return all_gather_into(t, shard, self.pg, async_op=True)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def train(self, n_steps: int, phased: Optional[bool] = None) -> torch.Tensor:
        """n_steps CQL steps; returns the per-step (rank-local share of the) loss as a device tensor -- no host sync
        inside the loop (cf. loss.item() per step at replay/models/base_torch_rec.py:39)."""
        losses = torch.zeros(max(n_steps, 1), dtype=torch.float32, device=self.device)
        # chunks keep the host at most a few thousand launches ahead of the device
        for lo in range(0, n_steps, 64):
            self.train_steps(min(64, n_steps - lo), losses[lo:], phased)
        if self.world > 1 and n_steps > 0:
            import torch.distributed as dist
            dist.all_reduce(losses, op=dist.ReduceOp.SUM, group=self.pg)
        return losses[:n_steps]

    def set_lr(self, lr: float) -> None:
        self.hyper.lr = float(lr)
        if self._ctx is not None:
            self._ctx.lr = float(lr)

    def eval_loss(self, offsets, items, rewards, n_batches: int, seed: int = 12345) -> float:
        """Mean CQL loss over n_batches sampled batches of ANOTHER log (validation), parameters untouched: the forward
        phase only (role of TorchRecommender._run_validation, replay/models/base_torch_rec.py:41-55)."""
        def dev(x, dt):
            t = torch.as_tensor(np.ascontiguousarray(x) if not torch.is_tensor(x) else x)
            return t.to(device=self.device, dtype=dt).contiguous()
        o, it, rw = dev(offsets, torch.int64), dev(items, torch.int32), dev(rewards, torch.float32)
        if o.numel() < 2 or int(o[-1]) != it.numel() or it.numel() == 0:
            raise ValueError("inconsistent or empty validation CSR")
        base = self._train_ctx()
        c = N.TrainCtx()
        C.memmove(C.byref(c), C.byref(base), C.sizeof(N.TrainCtx))
        c.offsets, c.items, c.rewards, c.n_users = _ptr(o), _ptr(it), _ptr(rw), o.numel() - 1
        c.seed, c.world, c.rank = int(seed), 1, 0
        losses = torch.zeros(max(n_batches, 1), dtype=torch.float32, device=self.device)
        for i in range(n_batches):
            N.check(self.lib.cqlrec_train_step_forward(C.byref(c), i, _ptr(losses[i:i + 1]), _stream()), "eval forward")
        # The forward phase also sorts the pairs of a backward that never comes here, on a library stream that only the
        # backward joins: those kernels read THIS log (o, it), which dies with this frame -- the whole device must be idle
        # before it does (a sync of the current stream alone let them read freed memory: a GPU fault, timing permitting).
        torch.cuda.synchronize(self.device)
        return float(losses[:n_batches].mean().item()) if n_batches else float("nan")

    def views(self, step: Optional[int] = None) -> Dict[str, torch.Tensor]:
        """Intermediates of step `step` (default: the current one, i.e. the last forward_backward before its
        apply_update) -- tests / debugging.  Copies, synchronises."""
        c = self._train_ctx()
        v = N.TrainViews()
        N.check(self.lib.cqlrec_train_views_get(C.byref(c), self.step if step is None else int(step), C.byref(v)),
                "train_views_get")
        B, d = self.hyper.batch, self.hyper.d
        torch.cuda.synchronize(self.device)
        ws = self._ws
        base = ws.data_ptr()

        def view(name, dtype, shape):
            nbytes = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
            off = getattr(v, name) - base
            return ws[off: off + nbytes].view(dtype).view(*shape).clone()
        out = {n: view(n, torch.int32, (B,)) for n in ("users", "tpos", "act", "a_star")}
        out.update({n: view(n, torch.float32, (B,)) for n in ("rew", "done", "q_a", "lse", "q_targ", "y", "coef")})
        out.update({n: view(n, torch.float32, (B, d)) for n in ("dH", "dh0", "h0_s")})
        out.update({n: view(n, torch.bfloat16, (B, d)) for n in ("hb_s", "hb_sn", "hb_tn")})
        return out

    # ------------------------------------------------------------------ inference
    def encode(self, offsets: torch.Tensor, items: torch.Tensor, users: torch.Tensor,
               ends: Optional[torch.Tensor] = None, end_delta: int = 0, use_target: bool = False) -> torch.Tensor:
        """bf16 state vectors h for (user, end) states; ends=None -> the user's whole history (predict-time state)."""
        h, lay = self.hyper, self.layout
        n = users.numel()
        flat_b = self.target_b if use_target else self.theta_b
        flat = self.target if use_target else self.theta
        h0b = torch.empty((n, h.d), dtype=torch.bfloat16, device=self.device)
        zb = torch.empty_like(h0b)
        hb = torch.empty_like(h0b)
        s = _stream()
        eb = flat_b.data_ptr()
        N.check(self.lib.cqlrec_gather_pool_fwd(eb + 2 * lay.off_E_in, _ptr(offsets), _ptr(items), _ptr(users), _ptr(ends),
                                                end_delta, n, h.window, h.d, None, _ptr(h0b), None, s), "gather_pool_fwd")
        fp = flat.data_ptr()
        # two launches here, not cqlrec_encoder_fwd (same bits): in the predict pass these run on a side stream next to the
        # seen-bitmap builder, and the one-launch form (8.5 KB of LDS per 32 rows) slowed that builder by 25 % (measured)
        N.check(self.lib.cqlrec_linear_bf16(_ptr(h0b), eb + 2 * lay.off_W1, fp + 4 * lay.off_b1, n, h.d, 1, None,
                                            _ptr(zb), s), "linear_bf16")
        N.check(self.lib.cqlrec_linear_bf16(_ptr(zb), eb + 2 * lay.off_W2, fp + 4 * lay.off_b2, n, h.d, 0, None,
                                            _ptr(hb), s), "linear_bf16")
        return hb

    def score_topk(self, hb: torch.Tensor, k: int, cand_items: Optional[torch.Tensor] = None,
                   seen: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, seen_rows: Optional[torch.Tensor] = None,
                   chunk: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Top-k (score desc, item id asc) for the state vectors hb.  cand_items: ascending int32 global ids (None =
        whole catalog).  seen = (offsets int64, ascending item ids int32) CSR; seen_rows maps hb rows to CSR rows.
        hb may also be a pair (n, fn) with fn(lo, hi) -> bf16 state vectors of rows [lo, hi): the vectors of a chunk are
        then produced while the chunk's seen bitmap is built on a side stream (encode_topk)."""
        h, lay = self.hyper, self.layout
        if isinstance(hb, tuple):
            n, hb = int(hb[0]), hb[1]
        else:
            n = hb.shape[0]
        if k > self.MAX_FUSED_K:
            return self._score_topk_large_k(hb, k, cand_items, seen, seen_rows, chunk)
        eb, fp = self.theta_b.data_ptr(), self.theta.data_ptr()
        if cand_items is None:
            E_ptr, b_ptr, n_cand, ids_ptr = eb + 2 * lay.off_E_out, fp + 4 * lay.off_b_out, self.n_items, None
            keep = ()
        else:  # compact the candidate rows once (plumbing: a row gather)
            ci = cand_items.to(device=self.device, dtype=torch.int64)
            E_sub = self.segment(self.theta_b, "E_out").index_select(0, ci).contiguous()
            b_sub = self.segment(self.theta, "b_out").index_select(0, ci).contiguous()
            ids = ci.to(torch.int32).contiguous()
            E_ptr, b_ptr, n_cand, ids_ptr = E_sub.data_ptr(), b_sub.data_ptr(), ids.numel(), ids.data_ptr()
            keep = (E_sub, b_sub, ids)
        out_idx = torch.empty((n, k), dtype=torch.int32, device=self.device)
        out_val = torch.empty((n, k), dtype=torch.float32, device=self.device)
        out_cnt = torch.empty((n,), dtype=torch.int32, device=self.device)
        if n == 0:
            return out_idx, out_val, out_cnt
        if seen is not None and seen_rows is None:
            seen_rows = torch.arange(n, dtype=torch.int32, device=self.device)
        if chunk is None:
            # users per launch of the scoring kernel: 131 072 (256 blocks of 512 users, qtopk4_kernel, the catalogue in
            # one slice) where that kernel applies, else 65 536 (256 blocks of 256 users, qtopk2_kernel)
            chunk = 131072 if (h.d == 128 and k <= 16 and cand_items is None and n >= 512 * 160) else 65536
        chunk = max(1, min(chunk, n))
        if seen is not None:    # the seen bitmap of a chunk (users x items bits) stays under 4 GiB
            chunk = min(chunk, max(4096, int((4 << 30) // max(1, n_cand // 8)) // 256 * 256))
        ws_bytes = int(self.lib.cqlrec_topk_ws_bytes(chunk, n_cand, h.d, k))
        main = torch.cuda.current_stream()
        s = main.cuda_stream
        hb_fn = hb if callable(hb) else None
        n_chunks = (n + chunk - 1) // chunk

        def call(ws_, lo, hi, hptr, phase, stream):
            rows_ptr = None if seen_rows is None else seen_rows.data_ptr() + 4 * lo
            N.check(self.lib.cqlrec_score_topk_phase(
                hptr, hi - lo, E_ptr, b_ptr, n_cand, h.d, ids_ptr,
                None if seen is None else _ptr(seen[0]), None if seen is None else _ptr(seen[1]), rows_ptr, k,
                _ptr(ws_), ws_bytes, out_idx.data_ptr() + 4 * k * lo, out_val.data_ptr() + 4 * k * lo,
                out_cnt.data_ptr() + 4 * lo, phase, stream), "score_topk")
        if hb_fn is None:
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
            for lo in range(0, n, chunk):
                hi = min(n, lo + chunk)
                call(ws, lo, hi, hb.data_ptr() + 2 * h.d * lo, N.TOPK_ALL, s)
        else:
            # The state vectors of chunk i+1 (window gather + encoder: small kernels, no LDS to speak of) are produced on a
            # side stream while chunk i is scored; the seen bitmap (820 MB of writes per 65 536 users x 100 000 items, and
            # a 48 KiB LDS tile per block that cannot be resident beside the scoring kernel anyway) is built on this
            # stream, in front of the scoring kernel that reads it.  (Also measured: bitmap AND states under the scoring
            # of the chunk before -- CQLREC_TOPK_SEEN_BESIDE, two workspaces: the HBM-bound bitmap slows the scoring
            # kernel by what it would cost alone; bitmap beside the encoder only: 2.19 ms per chunk, this order 2.0.)
            if self._topk_side is None:
                with torch.cuda.device(self.device):
                    self._topk_side = N.aux_stream(1)
            side = self._topk_side
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
            bounds = [(lo, min(n, lo + chunk)) for lo in range(0, n, chunk)]

            def encode_on_side(lo, hi):
                with torch.cuda.stream(side):
                    hb_c = hb_fn(lo, hi)
                    ev = torch.cuda.Event()
                    ev.record(side)
                return hb_c, ev
            # The side stream runs ahead unfenced.  Measured against CQL_TOPK_PIPE=fenced (the states of chunk i+1 produced
            # strictly in the window between two scoring kernels, beside the bitmap of chunk i, and nothing beside a
            # scoring kernel): 31.6 against 29.8 ms per pass over 1 M users -- the scoring kernel is no faster alone
            # (3.09 ms per 131 072 users either way) and the window grows by what the encoder no longer hides.
            fenced = os.environ.get("CQL_TOPK_PIPE", "free") == "fenced"
            side.wait_stream(main)
            nxt = encode_on_side(*bounds[0])
            for i, (lo, hi) in enumerate(bounds):
                hb_c, ready = nxt
                if i + 1 < len(bounds):
                    nxt = encode_on_side(*bounds[i + 1])
                call(ws, lo, hi, None, N.TOPK_SEEN, s)
                main.wait_event(ready)
                if fenced and i + 1 < len(bounds):
                    main.wait_event(nxt[1])
                call(ws, lo, hi, hb_c.data_ptr(), N.TOPK_SCORE, s)
                if fenced:
                    side.wait_stream(main)
                hb_c.record_stream(main)
        del keep
        return out_idx, out_val, out_cnt

    MAX_FUSED_K = 2048

    def encode_topk(self, offsets: torch.Tensor, items: torch.Tensor, users: torch.Tensor, k: int,
                    cand_items: Optional[torch.Tensor] = None,
                    seen: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, seen_rows: Optional[torch.Tensor] = None,
                    chunk: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """encode(offsets, items, users) + score_topk in one pass (the predict path, S7): the state vectors of chunk i+1
        are produced on a side stream while chunk i is scored."""
        if seen is not None and seen_rows is None:
            seen_rows = users.to(device=self.device, dtype=torch.int32)
        if k > self.MAX_FUSED_K:
            return self.score_topk(self.encode(offsets, items, users), k, cand_items, seen, seen_rows, chunk)
        return self.score_topk((users.numel(), lambda lo, hi: self.encode(offsets, items, users[lo:hi])), k, cand_items,
                               seen, seen_rows, chunk)

    def _score_topk_large_k(self, hb, k, cand_items, seen, seen_rows, chunk):
        """k beyond the fused kernel's limit (full-ranking requests): rank every candidate part of <= MAX_FUSED_K items
        completely with the fused kernel, then merge the parts per user (plumbing; ordering rule unchanged)."""
        n = hb.shape[0]
        ci = (torch.arange(self.n_items, device=self.device) if cand_items is None
              else cand_items.to(self.device)).to(torch.int64)
        parts_i, parts_v = [], []
        for lo in range(0, ci.numel(), self.MAX_FUSED_K):
            part = ci[lo: lo + self.MAX_FUSED_K]
            i_, v_, _ = self.score_topk(hb, int(part.numel()), part, seen, seen_rows, chunk)
            parts_i.append(i_)
            parts_v.append(v_)
        idx, val = torch.cat(parts_i, 1), torch.cat(parts_v, 1)
        # (score desc, item id asc); padding entries (-1, -inf) sink to the end
        key_id = torch.where(idx >= 0, idx, torch.full_like(idx, 2**31 - 1)).to(torch.int64)
        o1 = torch.argsort(key_id, dim=1, stable=True)
        o2 = torch.argsort(val.gather(1, o1), dim=1, descending=True, stable=True)
        order = o1.gather(1, o2)[:, :k]
        idx, val = idx.gather(1, order), val.gather(1, order)
        if idx.shape[1] < k:
            pad = k - idx.shape[1]
            idx = torch.nn.functional.pad(idx, (0, pad), value=-1)
            val = torch.nn.functional.pad(val, (0, pad), value=float("-inf"))
        return idx.contiguous(), val.contiguous(), (idx >= 0).sum(1).to(torch.int32)

    def pair_scores(self, hb: torch.Tensor, item_ids: torch.Tensor) -> torch.Tensor:
        """relevance of (hb[i], item_ids[i]) pairs (a11)."""
        h, lay = self.hyper, self.layout
        out = torch.empty((hb.shape[0],), dtype=torch.float32, device=self.device)
        N.check(self.lib.cqlrec_gather_dot(_ptr(hb), self.theta_b.data_ptr() + 2 * lay.off_E_out,
                                           self.theta.data_ptr() + 4 * lay.off_b_out, _ptr(item_ids), hb.shape[0], h.d,
                                           _ptr(out), _stream()), "gather_dot")
        return out
