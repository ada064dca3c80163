# GPU box: what a kernel on a second stream costs a chain of small dependent kernels (tools/probe_overlap.hip)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/probe_overlap
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWITH_LIB -Iinclude tools/probe_overlap.hip -o /tmp/probe_overlap -Lcamradepth_amd -l:libcamradepth_hip.so -Wl,-rpath,$GRAFT_REPO_ROOT/camradepth_amd 2>/dev/null || exit 1
timeout 300 /tmp/probe_overlap | tee gpurun_out/probe_overlap/result.txt
