#!/bin/bash
# Round-3 evidence run on the GPU box (from the repo root): bench lines of every config, rocprofv3 kernel stats of the
# default bench command, PMC passes over the d = 128 and d = 256 Q-head kernels.  Everything lands under gpurun_out/r03/.
set -o pipefail
root=$(pwd)
out=gpurun_out/r03
mkdir -p $out
export TMPDIR=/tmp
echo "[r03] bench cfg3 (driver flags and defaults)"
python bench.py --steps 20 --warmup 5 > $out/bench_cfg3_20_5.json 2> $out/bench_cfg3_20_5.err || exit 1
python bench.py > $out/bench_cfg3.json 2> $out/bench_cfg3.err || exit 1
echo "[r03] bench other configs"
for c in cfg1 cfg2 cfg5shard; do
  python bench.py --config $c --steps 50 --warmup 5 > $out/bench_$c.json 2> $out/bench_$c.err || exit 1
done
python bench.py --batch 16384 --steps 50 --warmup 10 --no-topk --no-cpu-baseline > $out/bench_cfg3_B16384.json 2> $out/bench_B16384.err || exit 1
echo "[r03] rocprofv3 kernel stats of the bench command"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/kt -- python3 $root/bench.py --steps 50 --warmup 10 --no-cpu-baseline) > $out/bench_under_rocprof.json 2> $out/kt.err || { tail -5 $out/kt.err; exit 1; }
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/kt
echo "[r03] PMC passes, d = 128 (cfg3 shape)"
tools/pmc_passes.sh $out/pmc128 "--modes argmax,fused,bwd --reps 5" sq1 sq2 fetch write > $out/pmc_qhead_d128.txt 2>&1 || { tail -5 $out/pmc_qhead_d128.txt; exit 1; }
echo "[r03] PMC passes, d = 256 (cfg5 per-GPU shape)"
tools/pmc_passes.sh $out/pmc256 "--items 1000000 --d 256 --modes argmax,fused,bwd --reps 2" sq1 sq2 fetch write > $out/pmc_qhead_d256.txt 2>&1 || { tail -5 $out/pmc_qhead_d256.txt; exit 1; }
echo "[r03] PMC passes, top-K scoring kernel (131 072 users x 100 000 items, qtopk4_kernel)"
tools/pmc_passes.sh $out/pmctk "--modes topk --topk-users 131072 --reps 3" sq1 sq2 > $out/pmc_topk.txt 2>&1 || { tail -5 $out/pmc_topk.txt; exit 1; }
rm -rf $out/pmc128 $out/pmc256 $out/pmctk
echo "[r03] done"
