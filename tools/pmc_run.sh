# SQ / LDS / cache counters of selected kernels of a bench.py run, collected on the GPU box:
#     gpurun --timeout 900 -- bash tools/pmc_run.sh <tag> <kernel name substring> <bench.py arguments...>
# e.g.  bash tools/pmc_run.sh wide64 k_dec_attn --brief --no-graph --fill prefill --steps 4 --warmup 1 ... (see below)
# Separate passes (8 SQ slots per pass); --kernel-trace only beside --pmc; eager launches (--no-graph) so that every dispatch
# is counted.  Summary -> gpurun_out/pmc_<tag>.txt (per-dispatch means).
TAG=$1; shift
SUB=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( while sleep 40; do echo tick; done ) & HB=$!
i=0
for SET in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
           "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "FETCH_SIZE" ; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- python3 $R/bench.py "$@" > /dev/null 2> $OUT/p$i.err
  echo "pass $i rc=$?"
done
kill $HB
python3 - $OUT "$SUB" <<'PY' > $R/gpurun_out/pmc_$TAG.txt
import csv, glob, os, sys
from collections import defaultdict
d, sub = sys.argv[1], sys.argv[2]
for p in sorted(os.listdir(d)):
    if not os.path.isdir(os.path.join(d, p)): continue
    agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
    for f in glob.glob(os.path.join(d, p, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:72]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k in sorted(agg):
        if sub in k:
            n = len(cnt[k])
            print(f"{k:72s} dispatches {n:5d} " + " ".join(f"{c}={v / n:.0f}" for c, v in sorted(agg[k].items())))
PY
cat $R/gpurun_out/pmc_$TAG.txt
rm -rf $OUT/p*/
