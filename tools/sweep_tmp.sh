for c in 128 256 384 512 768 1024; do
  echo "W3_WGS=$c: $(CRD_W3_WGS=$c python bench.py --no-cpu-baseline --no-roofline --steps 20 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')"
done
