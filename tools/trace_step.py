#!/usr/bin/env python3
"""Print the kernel timeline of one pipelined train step from a rocprofv3 --kernel-trace run (rocpd database).

    rocprofv3 --kernel-trace -d gpurun_out/tr -o run -- python3 tools/phase_timing.py --trace-only
    python tools/trace_step.py gpurun_out/tr/run_results.db [steps_from_end]
"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    rows = list(c.execute("select name,start,end,queue_id,stream_id from kernels order by start"))
    idx = [i for i, r in enumerate(rows) if "sample_kernel" in r[0]]
    k = len(idx) - (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
    a, b = idx[k], idx[k + 1]
    t0 = rows[a][1]
    lo = a
    while lo > 0 and rows[lo - 1][2] > t0:
        lo -= 1
    nsort = 0
    for r in rows[lo:b + 14]:
        n = r[0].split("(")[0]
        if "rocprim" in n:
            nsort += 1
            continue
        print(f"{(r[1] - t0) / 1e3:8.1f} {(r[2] - t0) / 1e3:8.1f} {(r[2] - r[1]) / 1e3:7.1f} q={r[3]} s={r[4]} {n[:60]}")
    print("step period (sample to sample): %.1f us; %d rocprim launches hidden" % ((rows[b][1] - t0) / 1e3, nsort))
    per = [(rows[idx[i + 1]][1] - rows[idx[i]][1]) / 1e3 for i in range(len(idx) - 1)]
    per.sort()
    print("median period over the run: %.1f us" % per[len(per) // 2])


if __name__ == "__main__":
    main()
