#!/bin/bash
# Round 5, one GPU call: the register epilogue of the 64 x 64 igemm tiles on / off in the C2 step, interleaved runs.
O=gpurun_out/r5; mkdir -p $O
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('ms_per_step', d.get('ms_per_forward')))"; }
for i in 1 2 3 4; do
  a=$(python bench.py --no-cpu-baseline --no-roofline --steps 60 --tune-rege 1 2>/dev/null | ms)
  b=$(python bench.py --no-cpu-baseline --no-roofline --steps 60 --tune-rege 0 2>/dev/null | ms)
  echo "run $i: train step, register epilogue on $a / off $b ms"
done 2>&1 | tee $O/ab_rege.txt
