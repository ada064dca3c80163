cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02c
for i in 1 2; do
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline | cut -c1-160
done
cp camradepth_amd/libcamradepth_hip.so /tmp/keep.so; cp camradepth_amd/libalt_wdma.so camradepth_amd/libcamradepth_hip.so
echo "-- weights by LDS-DMA"
for i in 1 2; do
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline | cut -c1-160
done
cp /tmp/keep.so camradepth_amd/libcamradepth_hip.so
