cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/inf_b1; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o fwd -- python3 bench.py --inference --batch 1 --height 416 --width 800 --steps 30 > $O/bench.json 2> $O/bench.err
python3 tools/rocprof_forward.py $(ls $O/trace/*.db | head -1) > $O/forward_only_kernels.txt 2>&1
python3 tools/rocprof_timeline.py $(ls $O/trace/*.db | head -1) $O/timeline.txt > /dev/null 2>&1
rm -rf $O/trace; head -50 $O/forward_only_kernels.txt
