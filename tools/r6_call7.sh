#!/bin/bash
O=gpurun_out/r6; mkdir -p $O
( time timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline ) > $O/bench_excess.json 2> $O/bench_excess.err; tail -4 $O/bench_excess.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6/bench_excess.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d.get("excess_ms"), d["roofline"]["frac"], d["floor_ms"], d["step_over_floor"], {k: v["launches"] for k, v in d["floor_budget"]["phases"].items()})
PY
