cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02b
( time timeout 3000 python -m pytest tests/test_gpu_gnconv.py -m gpu -q -s 2>&1 ) > gpurun_out/r02b/tests_gn.log 2>&1
( time CRD_GN_CONV=1 timeout 3000 python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py tests/test_gpu_data.py -m gpu -q -s 2>&1 ) > gpurun_out/r02b/tests.log 2>&1
CRD_GN_CONV=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02b/bench_c2.json 2> gpurun_out/r02b/bench_c2.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r02b/bench_c2_nogn.json 2> gpurun_out/r02b/bench_c2_nogn.err
CRD_GN_CONV=1 timeout 600 python tools/profile_ops.py 8 > gpurun_out/r02b/ops_table.txt 2>&1
grep -n "passed\|failed" gpurun_out/r02b/tests_gn.log gpurun_out/r02b/tests.log | tail -3; cut -c1-300 gpurun_out/r02b/bench_c2.json; cut -c1-300 gpurun_out/r02b/bench_c2_nogn.json
python - <<'PY'
import json, collections
d=json.load(open('gpurun_out/ops_all.json'))
by=collections.defaultdict(list)
for ph,i,name,shape,ms in d:
    if name=='crd_gn_conv': by[shape].append(ms)
for k,v in by.items(): print(f"{k:60s} x{len(v):3d} avg {sum(v)/len(v)*1e3:7.1f} us")
PY
