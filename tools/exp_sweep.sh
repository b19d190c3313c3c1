for fp in 1 0 1 0; do echo "== FWDPIPE=$fp"; CQL_QS_FWDPIPE=$fp python tools/qhead_microbench.py --modes lse,argmax,topk --reps 10 2>&1 | grep -E "qhead_(lse|argmax)|topk_tilemax"; done
