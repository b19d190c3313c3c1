#!/bin/bash
# kernel timeline of the training step (bench.py, no top-K pass): gpurun_out/st_tl/*kernel_trace.csv
export TMPDIR=/tmp
rm -rf gpurun_out/st_tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/st_tl -- python3 bench.py --steps 40 --warmup 20 --no-cpu-baseline --no-prof --no-topk "$@" > gpurun_out/st_tl.log 2>&1
ls gpurun_out/st_tl/*/ | head
