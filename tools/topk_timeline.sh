#!/bin/bash
# kernel timeline of the predict pass (bench.py, few training steps): gpurun_out/tk_tl/*kernel_trace.csv
export TMPDIR=/tmp
rm -rf gpurun_out/tk_tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tk_tl -- python3 bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-prof --topk-chunk ${1:-131072} > gpurun_out/tk_tl.log 2>&1
ls gpurun_out/tk_tl/*/ | head
