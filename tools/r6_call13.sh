#!/bin/bash
O=gpurun_out/r6; mkdir -p $O
timeout 120 python -m pytest tests/test_gpu_ops.py -x -q -k "never_reads" 2>&1 | tail -2
PYTHONPATH=. timeout 900 python tools/exp_cumask.py 2>&1 | grep -v amdgpu.ids | tee $O/exp_cumask.txt
