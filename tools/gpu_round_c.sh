cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02c
( timeout 1800 python -m pytest tests/test_gpu_igemm.py -m gpu -q -k persistent 2>&1 | tail -3 ) > gpurun_out/r02c/tests_p.log 2>&1
tail -2 gpurun_out/r02c/tests_p.log
for cfg in "304 128 0 0" "240 64 0 0" "144 96 0 0" "128 304 1 0" "128 304 1 1" "64 240 1 1" "96 144 1 1"; do
  set -- $cfg
  for w in 4 8; do echo -n "waves=$w acc=$4 "; ACC=$4 CRD_CONV3P_WAVES=$w CIN=$1 COUT=$2 timeout 120 python tools/bench_conv.py $3 20 $([ $3 = 1 ] && echo nostats) 2>&1 | tail -1; done
done
