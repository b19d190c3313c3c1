#!/usr/bin/env python3
"""One steady-state training step out of a `rocprofv3 --kernel-trace --output-format csv` run: every kernel between two
consecutive td_loss launches, start / end in microseconds after the first one.

    python tools/step_timeline.py <kernel_trace.csv> [--skip 100]
"""
import argparse
import csv
import re


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("rocprim::ROCPRIM_400001_NS::detail::", "rp::").replace("rocprim::ROCPRIM_400200_NS::detail::", "rp::")
    return name[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--skip", type=int, default=100, help="td_loss launches to skip (warm-up)")
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", ""))))
    rows.sort()
    td = [i for i, r in enumerate(rows) if r[2].startswith("td_loss")]
    i0, i1 = td[a.skip], td[a.skip + 1]
    t0 = rows[i0][0]
    print(f"step = {(rows[i1][0] - t0) / 1e3:.1f} us")
    for s, e, n, q in rows[i0:i1 + 1]:
        print(f"{(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  q{q:>3}  {short(n)}")


if __name__ == "__main__":
    main()
