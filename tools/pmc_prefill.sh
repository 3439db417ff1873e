# SQ / LDS / cache counters of the prompt-processing kernels (2048-id q4 prefill), collected on the GPU box:
#     gpurun --timeout 900 -- bash tools/pmc_prefill.sh [tag]
# separate passes (8 SQ slots per pass), --kernel-trace only beside --pmc; summary -> gpurun_out/pmc_prefill_<tag>.txt
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_prefill_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( while sleep 40; do echo tick; done ) & HB=$!
ARGS="--no-cpu-baseline --no-graph --streams 0 --wide-streams 0 --generate 0 --serve 0 --steps 2 --warmup 1 --fill prefill --prefill 2048"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/p1 -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/p1.err
echo "p1 rc=$?"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $OUT/p2 -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/p2.err
echo "p2 rc=$?"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/p3 -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/p3.err
echo "p3 rc=$?"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/p4 -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/p4.err
echo "p4 rc=$?"
kill $HB
python3 - $OUT <<'PY' > $R/gpurun_out/pmc_prefill_$TAG.txt
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
for p in ("p1", "p2", "p3", "p4"):
    agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
    for f in glob.glob(os.path.join(d, p, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:64]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k in sorted(agg):
        if any(s in k for s in ("k_matmul_mfma", "k_attn_tiled", "k_act_to_f16")):
            n = len(cnt[k])
            print(f"{k:64s} dispatches {n:5d} " + " ".join(f"{c}={v / n:.0f}" for c, v in sorted(agg[k].items())))
PY
cat $R/gpurun_out/pmc_prefill_$TAG.txt
rm -rf $OUT        # (raw counter files: far beyond what gpurun copies back)
