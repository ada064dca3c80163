#!/bin/bash
# Round 5, one GPU call: (1) the narrow streaming pointwise kernel on / off in the C2 step and the B = 8 inference forward, three runs each,
# interleaved; (2) the late stream's workgroup budget re-tuned on the round-5 kernels (developer switch CRD_W3_LATE_WGS, Python side).
O=gpurun_out/r5; mkdir -p $O
ms() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('ms_per_step', d.get('ms_per_forward')))"; }
{
for i in 1 2 3; do
  a=$(python bench.py --no-cpu-baseline --no-roofline --steps 60 2>/dev/null | ms)
  b=$(python bench.py --no-cpu-baseline --no-roofline --steps 60 --tune-narrow 0 2>/dev/null | ms)
  c=$(python bench.py --inference --batch 8 --steps 30 2>/dev/null | ms)
  d=$(python bench.py --inference --batch 8 --steps 30 --tune-narrow 0 2>/dev/null | ms)
  echo "run $i: train step narrow on $a / off $b ms; inference forward B=8 narrow on $c / off $d ms"
done
for w in 128 144 160 192 224; do
  a=$(CRD_DEV_SWITCHES=1 CRD_W3_LATE_WGS=$w python bench.py --no-cpu-baseline --no-roofline --steps 60 2>/dev/null | ms)
  echo "late-stream workgroup budget $w: $a ms per step"
done
} 2>&1 | tee $O/sweep_round5.txt
