for tb in 768 512 1024 1536 768; do echo "== BLOCKS=$tb"; CQL_QS_BLOCKS=$tb python tools/qhead_microbench.py --modes lse,argmax,bwd --reps 10 2>&1 | grep -E "qhead_(lse|argmax|bwd)"; done
