#!/bin/bash
O=gpurun_out/final6; mkdir -p $O
timeout 600 python bench.py --batch 4 --height 928 --width 1600 --freeze-seg --variant supervised_seg --steps 10 --warmup 3 --no-cpu-baseline --no-excess > $O/bench_c4.json 2> $O/bench_c4.err; cut -c1-250 $O/bench_c4.json; tail -2 $O/bench_c4.err | cut -c1-200
timeout 300 python bench.py --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-excess 2>&1 | tail -1 | cut -c1-200
