#!/usr/bin/env python3
"""Micro-benchmark of the streaming Q-head kernels (phase "qhead_lse" counts both the plain LSE launch of mode `lse`
and the fused LSE+dH launch of mode `fused`: run them in separate invocations to tell them apart) at a BASELINE shape (inputs resident in HBM, HIP-event timing
through the library's measurement hooks).  Used for A/B-ing kernel variants and as the rocprofv3 --pmc target.

    python tools/qhead_microbench.py [--batch 4096] [--items 100000] [--d 128] [--reps 20] [--modes lse,argmax,bwd,topk]
"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from replay_cql_amd import _native as N  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--items", type=int, default=100_000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--modes", default="lse,argmax,fused,bwd,topk")
    ap.add_argument("--topk-users", type=int, default=16384)
    a = ap.parse_args()
    lib = N.load()
    dev = "cuda:0"
    B, NI, d = a.batch, a.items, a.d
    g = torch.Generator(device=dev).manual_seed(0)
    H = (torch.randn(B, d, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    E = (torch.randn(NI, d, device=dev, generator=g) / d ** 0.5).to(torch.bfloat16)
    b = torch.randn(NI, device=dev, generator=g) * 0.1
    s = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(max(int(lib.cqlrec_qhead_ws_bytes(B, NI, d)), int(lib.cqlrec_qhead_bwd_ws_bytes(B, NI, d))),
                     dtype=torch.uint8, device=dev)
    lse = torch.empty(B, device=dev)
    nlse2 = torch.empty(B, device=dev)
    vmax = torch.empty(B, device=dev)
    imax = torch.empty(B, dtype=torch.int32, device=dev)
    act = torch.randint(0, NI, (B,), device=dev, dtype=torch.int32, generator=g)
    coef = torch.randn(B, device=dev, generator=g) * 1e-3
    dH = torch.empty(B, d, device=dev)
    gE = torch.empty(NI, d, device=dev)
    gb = torch.empty(NI, device=dev)
    modes = a.modes.split(",")

    def run_lse():
        N.check(lib.cqlrec_qhead_fwd(H.data_ptr(), B, E.data_ptr(), b.data_ptr(), NI, d, N.QHEAD_LSE, ws.data_ptr(),
                                     ws.numel(), lse.data_ptr(), None, nlse2.data_ptr(), s))

    def run_argmax():
        N.check(lib.cqlrec_qhead_fwd(H.data_ptr(), B, E.data_ptr(), b.data_ptr(), NI, d, N.QHEAD_ARGMAX, ws.data_ptr(),
                                     ws.numel(), vmax.data_ptr(), imax.data_ptr(), None, s))

    def run_bwd():
        N.check(lib.cqlrec_qhead_bwd(H.data_ptr(), nlse2.data_ptr(), coef.data_ptr(), act.data_ptr(), B, E.data_ptr(),
                                     b.data_ptr(), NI, d, 1.0 / B, ws.data_ptr(), ws.numel(), dH.data_ptr(), gE.data_ptr(),
                                     gb.data_ptr(), s))
    nu, k = a.topk_users, 10
    Hu = (torch.randn(nu, d, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    wsk = torch.empty(int(lib.cqlrec_topk_ws_bytes(nu, NI, d, k)), dtype=torch.uint8, device=dev)
    oi = torch.empty(nu, k, dtype=torch.int32, device=dev)
    ov = torch.empty(nu, k, device=dev)
    oc = torch.empty(nu, dtype=torch.int32, device=dev)

    def run_topk():
        N.check(lib.cqlrec_score_topk(Hu.data_ptr(), nu, E.data_ptr(), b.data_ptr(), NI, d, None, None, None, None, k,
                                      wsk.data_ptr(), wsk.numel(), oi.data_ptr(), ov.data_ptr(), oc.data_ptr(), s))
    wsf = torch.empty(int(lib.cqlrec_qhead_fused_ws_bytes(B, NI, d)), dtype=torch.uint8, device=dev)

    def run_fused():   # the training step's forward of branch A: logsumexp + softmax-weighted item sum in one pass
        N.check(lib.cqlrec_qhead_fwd_lse_dh(H.data_ptr(), B, E.data_ptr(), b.data_ptr(), NI, d, wsf.data_ptr(), wsf.numel(),
                                            lse.data_ptr(), nlse2.data_ptr(), s))
        N.check(lib.cqlrec_qhead_dh_finish(wsf.data_ptr(), B, NI, d, lse.data_ptr(), coef.data_ptr(), act.data_ptr(),
                                           E.data_ptr(), 1.0 / B, dH.data_ptr(), s))
    fns = {"lse": run_lse, "argmax": run_argmax, "fused": run_fused, "bwd": run_bwd, "topk": run_topk}
    run_lse()
    torch.cuda.synchronize()
    for m in modes:
        for _ in range(3):
            fns[m]()
    torch.cuda.synchronize()
    N.check(lib.cqlrec_prof_enable(1))
    for _ in range(a.reps):
        for m in modes:
            fns[m]()
    torch.cuda.synchronize()
    ph = N.prof_read()
    N.check(lib.cqlrec_prof_enable(0))
    flops = 2.0 * B * NI * d
    for p, (ms, n) in ph.items():
        if not n:
            continue
        avg = ms / n
        line = f"{p:14s} launches={n:4d} avg_ms={avg:8.4f}"
        if p in ("qhead_lse", "qhead_argmax", "qhead_bwd_dh", "qhead_bwd_de"):
            line += f"  algorithmic {flops / avg / 1e9:8.1f} TFLOP/s ({flops / avg / 1e9 / 2500 * 100:5.1f}% of 2.5 PF)"
        if p == "topk_tilemax":
            f2 = 2.0 * nu * NI * d
            line += f"  algorithmic {f2 / avg / 1e9:8.1f} TFLOP/s ({f2 / avg / 1e9 / 2500 * 100:5.1f}% of 2.5 PF)"
        print(line)


if __name__ == "__main__":
    main()
