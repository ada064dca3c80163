#!/bin/bash
# what did rounding the depthwise 3x3 weights to bf16 (round 4) buy?  A/B at FIXED kernels, where nothing is amplified
# (every Block alone on the oracle's input, tests/test_gpu_blocks.py) and end to end (the chaotic golden-weight forward, tests/test_gpu_model.py).
# Run on an MI355X box from the repository root; writes gpurun_out/r5/ab_dw_rounding.txt
O=gpurun_out/r5; mkdir -p $O
{
echo "== depthwise weights rounded to bf16 (default) =="
python -m pytest tests/test_gpu_blocks.py -q -s -k "False" 2>&1 | grep -E "worst block|passed|failed"
python -m pytest tests/test_gpu_model.py -q -s -k "256x416_matches_reference_golden or rmse_gap_with_golden" 2>&1 | grep -E "MEASURED|rel-L2|passed|failed" | head -20
echo "== depthwise weights un-rounded fp32 (CRD_DEV_SWITCHES=1 CRD_DW_F32=1) =="
CRD_DEV_SWITCHES=1 CRD_DW_F32=1 python -m pytest tests/test_gpu_blocks.py -q -s -k "False" 2>&1 | grep -E "worst block|passed|failed"
CRD_DEV_SWITCHES=1 CRD_DW_F32=1 python -m pytest tests/test_gpu_model.py -q -s -k "256x416_matches_reference_golden or rmse_gap_with_golden" 2>&1 | grep -E "MEASURED|rel-L2|passed|failed" | head -20
} > $O/ab_dw_rounding.txt 2>&1
cat $O/ab_dw_rounding.txt
