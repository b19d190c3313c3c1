#!/usr/bin/env python3
"""HBM-bound kernels on tables that do NOT fit the 256 MiB Infinity Cache (the bench's E_in table, 25.6 MB, does):
the window gather (a3), its sorted scatter backward (a3') and the fused Adam + Polyak + bf16-shadow update (a9).
Prints, per kernel, the HIP-event time and the ALGORITHMIC bytes; run under `rocprofv3 --pmc FETCH_SIZE`
and `--pmc WRITE_SIZE` (tools/pmc_passes.sh with PMC_TOOL=tools/hbm_microbench.py) for the measured traffic.

    python tools/hbm_microbench.py [--items 2000000] [--d 128] [--states 262144] [--L 50] [--dist zipf|uniform]
                                   [--params 268435456] [--reps 5] [--modes gather,gather_bwd,adam]
"""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from replay_cql_amd import _native as N  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=2_000_000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--states", type=int, default=262_144)
    ap.add_argument("--L", type=int, default=50)
    ap.add_argument("--dist", default="zipf", choices=["zipf", "uniform"])
    ap.add_argument("--params", type=int, default=256 * 1024 * 1024)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--modes", default="gather,gather_bwd,adam")
    a = ap.parse_args()
    lib = N.load()
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(0)
    out = {"config": vars(a)}

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps

    modes = a.modes.split(",")
    if "gather" in modes or "gather_bwd" in modes:
        NI, d, S, L = a.items, a.d, a.states, a.L
        E = (torch.randn(NI, d, device=dev, generator=g) / d ** 0.5).to(torch.bfloat16)
        if a.dist == "zipf":      # p(rank) ~ 1 / rank (the synthetic log's popularity law), ids shuffled over the table
            u = torch.rand(S * L, device=dev, generator=g)
            rank = torch.exp(u * torch.log(torch.tensor(float(NI), device=dev))).to(torch.int64).clamp_(1, NI) - 1
            perm = torch.randperm(NI, device=dev, generator=g)
            items = perm[rank].to(torch.int32)
        else:
            items = torch.randint(0, NI, (S * L,), device=dev, generator=g, dtype=torch.int32)
        offsets = torch.arange(S + 1, device=dev, dtype=torch.int64) * L
        users = torch.arange(S, device=dev, dtype=torch.int32)
        h0b = torch.empty(S, d, device=dev, dtype=torch.bfloat16)
        distinct = int(torch.unique(items).numel())
        if "gather" in modes:
            ms = timed(lambda: N.check(lib.cqlrec_gather_pool_fwd(E.data_ptr(), offsets.data_ptr(), items.data_ptr(),
                                                                  users.data_ptr(), None, 0, S, L, d, None,
                                                                  h0b.data_ptr(), None, s), "gather_pool_fwd"))
            alg = S * L * (2 * d + 4) + S * 2 * d          # gathered rows + indices in, bf16 state out
            out["gather_fwd"] = {"ms": ms, "algorithmic_bytes": alg, "algorithmic_GBps": alg / ms / 1e6,
                                 "table_bytes": NI * d * 2, "distinct_rows": distinct,
                                 "distinct_row_bytes": distinct * d * 2}
        if "gather_bwd" in modes:
            dh0 = torch.randn(S, d, device=dev, generator=g)
            gE = torch.zeros(NI, d, device=dev)
            wsb = int(lib.cqlrec_gather_pool_bwd_ws_bytes(S, L, d))
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)

            def bwd():
                N.check(lib.cqlrec_gather_pool_bwd_sorted(dh0.data_ptr(), offsets.data_ptr(), items.data_ptr(),
                                                          users.data_ptr(), None, 0, S, L, d, NI, ws.data_ptr(), wsb,
                                                          gE.data_ptr(), s), "gather_pool_bwd_sorted")
            ms = timed(bwd)
            # pairs written + sorted (8 B key/value, ~3 radix passes r+w), dh0 rows re-read per pair, distinct rows r+w
            alg = S * L * 4 + S * L * d * 4 + 2 * distinct * d * 4
            out["gather_bwd_sorted"] = {"ms": ms, "algorithmic_bytes": alg, "algorithmic_GBps": alg / ms / 1e6,
                                        "note": "g_E_in accumulates over the reps (not re-zeroed): timing only"}
        del E
    if "adam" in modes:
        P = a.params
        theta = torch.randn(P, device=dev, generator=g)
        grads = torch.randn(P, device=dev, generator=g) * 1e-3
        m = torch.zeros(P, device=dev)
        v = torch.zeros(P, device=dev)
        target = theta.clone()
        tb = torch.empty(P, device=dev, dtype=torch.bfloat16)
        gb = torch.empty(P, device=dev, dtype=torch.bfloat16)
        for zero in (0, 1):
            ms = timed(lambda: N.check(lib.cqlrec_adam_ema(theta.data_ptr(), grads.data_ptr(), m.data_ptr(), v.data_ptr(),
                                                           target.data_ptr(), tb.data_ptr(), gb.data_ptr(), P, 3e-4,
                                                           1.0, 0.9, 0.999, 1e-8, 0.005, zero, s), "adam_ema"))
            alg = P * (44 if zero else 40)     # r: g, theta, m, v, target; w: theta, m, v, target, 2 x bf16 (+ g zeroed)
            out["adam_zero%d" % zero] = {"ms": ms, "algorithmic_bytes": alg, "algorithmic_GBps": alg / ms / 1e6,
                                         "frac_of_8TBps": alg / ms / 1e6 / 8000.0}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
