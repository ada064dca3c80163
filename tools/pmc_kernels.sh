# GPU box: hardware-counter summary (one rocprofv3 --pmc pass per counter set, no tracing) of the two biggest matrix
# kernels at the benchmark's largest layer: 3x3 forward 304->128 (k_conv3x3p) and its weight gradient (k_wgrad3x3).
# Output: gpurun_out/pmc_kernels/summary.txt  (copied to profiles/ by hand)
cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_kernels; rm -rf $O; mkdir -p $O
SETS=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM"
 "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
 "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC"
 "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
 "TCC_REQ_sum TCC_BUSY_avr TCC_TAG_STALL_sum TCP_TAGRAM0_REQ_sum"
)
: > $O/summary.txt
for prog in "tools/bench_conv.py 0 3" "tools/bench_wgrad.py"; do
  echo "== $prog" >> $O/summary.txt
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    timeout 240 rocprofv3 --pmc $set -d $O/p$i -o p --output-format csv -- python3 $prog > $O/p$i.log 2>&1
    f=$(ls $O/p$i/*counter_collection.csv 2>/dev/null | head -1)
    python3 - "$f" >> $O/summary.txt <<'PY'
import csv, sys, collections
if len(sys.argv) < 2 or not sys.argv[1]:
    print("   (no output)"); sys.exit()
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_conv3x3p" in k or "k_wgrad3x3" in k:
        name = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"   {k:34s} {c:40s} launches={len(v)} avg={sum(v)/len(v):.5g}")
PY
    rm -rf $O/p$i
  done
done
cat $O/summary.txt
