#!/bin/bash
O=gpurun_out/r6; mkdir -p $O
timeout 1500 python -m pytest tests/ -q -m gpu -s > $O/gpu_tests_full.log 2>&1; tail -5 $O/gpu_tests_full.log
