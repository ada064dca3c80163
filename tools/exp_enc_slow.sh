#!/bin/bash
# Developer experiment: the per-process slow mode of the persistent encoder stage under the inference graph -- N processes per setting
# of CRD_ENC_PERSIST, ms per forward of each.
N=${1:-10}
ms() { python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_forward'])"; }
for i in $(seq $N); do
  a=$(CRD_ENC_PERSIST=auto python bench.py --inference --batch 8 --steps 30 2>/dev/null | ms)
  b=$(CRD_ENC_PERSIST=1 python bench.py --inference --batch 8 --steps 30 2>/dev/null | ms)
  c=$(CRD_ENC_PERSIST=0 python bench.py --inference --batch 8 --steps 30 2>/dev/null | ms)
  echo "auto $a   both $b   off $c"
done
