cd "$GRAFT_REPO_ROOT"
for r in 1 4096 16384 1073741824; do
 for b in 8 16; do
  t=$(CRD_GN_CONV_MAXROWS=$r timeout 600 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-roofline | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  i=$(CRD_GN_CONV_MAXROWS=$r timeout 600 python bench.py --batch $b --steps 30 --inference | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_forward'])")
  echo "MAXROWS=$r B=$b train $t ms  infer $i ms"
 done
done
