# GPU box: hardware-counter summary (one rocprofv3 --pmc pass per counter set, no tracing) of the ENCODER kernels inside an eager
# training iteration of the benchmark configuration (the "dependent-latency-bound" diagnosis needs counters
# under it) -- the small GEMMs, GroupNorm backward, depthwise, attention backward
# and the whole-iteration totals (MFMA busy share, HBM bytes).  Output: gpurun_out/pmc_encoder/summary.txt
cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_encoder; rm -rf $O; mkdir -p $O
SETS=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
 "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VMEM"
 "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_MISSES"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
: > $O/summary.txt
run_set() {   # $1 label, $2 env, $3.. program
  local label=$1 envs=$2; shift 2
  echo "== $label" >> $O/summary.txt
  local i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    env $envs true
    ( export $envs; timeout 300 rocprofv3 --pmc $set -d $O/p$i -o p --output-format csv -- python3 "$@" > $O/p$i.log 2>&1 )
    f=$(ls $O/p$i/*counter_collection.csv 2>/dev/null | head -1)
    python3 - "$f" "$set" >> $O/summary.txt <<'PY'
import csv, sys, collections
if len(sys.argv) < 2 or not sys.argv[1]:
    print("   (no output for", sys.argv[2] if len(sys.argv) > 2 else "?", ")"); sys.exit()
WANT = ("k_igemm<2, 2, 1, 1", "k_gngemm_reg", "k_gn_bwd_apply", "k_gn_bwd_reduce", "k_dwconv", "k_attn_scores_bwd", "k_attn_scores", "k_gnbwd_gemm", "k_attn_out_bwd",
        "k_conv3x3p", "k_wgrad3x3", "k_pw_narrow", "k_gn_pw_wide", "k_wgrad_grouped")
agg = collections.defaultdict(list)
tot = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    v = float(r["Counter_Value"])
    tot[r["Counter_Name"]] += v
    for w in WANT:
        if k.startswith(w):
            fam = w if w != "k_igemm<2, 2, 1, 1" else "k_igemm<2,2,1,1,*>"
            agg[(fam, r["Counter_Name"])].append(v)
for (k, c), v in sorted(agg.items()):
    print(f"   {k:24s} {c:34s} launches={len(v):5d} avg={sum(v)/len(v):14.6g} sum={sum(v):14.6g}")
for c, v in sorted(tot.items()):
    print(f"   {'(all kernels of the run)':24s} {c:34s} sum={v:14.6g}")
PY
    rm -rf $O/p$i
  done
}
run_set "eager training iterations x3 (per-launch encoder), base 8x7x256x416" "CRD_UNUSED=0" tools/run_forward.py 3 train
cat $O/summary.txt
