for v in "" 1 "" 1; do echo "== GENERIC=$v"; env ${v:+CQL_TOPK_GENERIC=1} python tools/qhead_microbench.py --modes topk --reps 10 2>&1 | grep -E "topk"; done
