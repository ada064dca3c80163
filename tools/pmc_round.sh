# GPU box: HBM traffic per kernel (two separate --pmc passes, as MI355X_MICROARCH.md prescribes) -> gpurun_out/pmc_round/traffic.json
cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_round; mkdir -p gpurun_out/pmc_round
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_round/f -o p -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-excess > gpurun_out/pmc_round/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_round/w -o p -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-excess > gpurun_out/pmc_round/w.log 2>&1
python3 tools/pmc_traffic.py $(ls gpurun_out/pmc_round/f/*.db | head -1) $(ls gpurun_out/pmc_round/w/*.db | head -1) gpurun_out/pmc_round/traffic.json | head -12
rm -rf gpurun_out/pmc_round/f gpurun_out/pmc_round/w
