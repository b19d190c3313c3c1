#!/bin/bash
# rocprofv3 counter passes over the Q-head micro-benchmark (one pass per counter set: the SQ block has 8 slots, the TCC
# counters FETCH_SIZE / WRITE_SIZE do not fit one pass together).  Usage, on the GPU box, from the repo root:
#   tools/pmc_passes.sh OUTDIR "microbench args" [passes...]       passes: sq1 sq2 tcc fetch write (default: sq1 sq2)
# PMC_TOOL=tools/topk_bench.py (or any script under the repo root) profiles that script instead of the micro-benchmark.
# Summarise with:  python tools/pmc_summary.py OUTDIR/*/*/*counter_collection.csv
set -e
out=$1; args=$2; shift 2
passes=${@:-sq1 sq2}
root=$(pwd)
export TMPDIR=/tmp
for p in $passes; do
  case $p in
    sq1) c="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE";;
    sq2) c="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE";;
    tcc) c="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum";;
    fetch) c="FETCH_SIZE";;
    write) c="WRITE_SIZE";;
  esac
  (cd /tmp && rocprofv3 --pmc $c --output-format csv -d $root/$out/$p -- python3 $root/${PMC_TOOL:-tools/qhead_microbench.py} $args) > $out.$p.log 2>&1 || { tail -5 $out.$p.log; exit 1; }
done
python3 tools/pmc_summary.py $out/*/*/*counter_collection.csv
