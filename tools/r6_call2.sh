#!/bin/bash
# round 6, call 2: warm per-launch times of the whole backward chain with / without the GroupNorm-backward fusions
O=gpurun_out/r6; mkdir -p $O
CRD_DEV_SWITCHES=1 CRD_GNB_FC1=15 CRD_GNB_SR=15 PYTHONPATH=. python tools/chain_table.py bwd 0 2000 > $O/chain_bwd_gnb.log 2>/dev/null
CRD_DEV_SWITCHES=1 CRD_GNB_FC1=0 CRD_GNB_SR=0 PYTHONPATH=. python tools/chain_table.py bwd 0 2000 > $O/chain_bwd_base.log 2>/dev/null
tail -n 1 $O/chain_bwd_gnb.log; tail -n 1 $O/chain_bwd_base.log
