cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --list-avail > gpurun_out/avail.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d gpurun_out/pmc_conv$i -o p --output-format csv -- python3 tools/bench_conv.py 0 3 > gpurun_out/pmc_conv$i.log 2>&1
  f=$(ls gpurun_out/pmc_conv$i/*counter_collection.csv 2>/dev/null | head -1)
  echo "set $i: $f"
  python3 - "$f" <<'PY'
import csv, sys, collections
if len(sys.argv) < 2 or not sys.argv[1]:
    sys.exit()
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_conv3x3" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"   {k:36s} n={len(v)} avg={sum(v)/len(v):.4g}")
PY
done
