for c in 128 256 512 1024 2048; do
  echo "GN_CAP_R=$c: $(CRD_GN_CAP_R=$c python bench.py --no-cpu-baseline --no-roofline --steps 20 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')"
done
