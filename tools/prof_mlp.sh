# GPU box: k_mlp_fwd phase stamps
cd "$GRAFT_REPO_ROOT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCRD_MLP_PROF -c camradepth_amd/csrc/mlp_fused.hip -o /tmp/mlp_prof.o
cp camradepth_amd/libcamradepth_hip.so /tmp/lib_backup.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o camradepth_amd/libcamradepth_hip.so $(ls camradepth_amd/csrc/build/*.o | grep -v mlp_fused.o) /tmp/mlp_prof.o
python3 tools/prof_mlp.py
cp /tmp/lib_backup.so camradepth_amd/libcamradepth_hip.so
