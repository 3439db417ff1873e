# Counters behind the roofline objects of bench.py's SECONDARY legs (the batch-1 leg's come from tools/collect_profiles.sh):
#     gpurun --timeout 1200 -- bash tools/collect_counters.sh [tag]
#   lanes    rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, eager launches) of the 256-sequence step (two
#            lanes of 128 rows): HBM bytes per step over every k_dec_* dispatch, gfx950 correction applied
#   prefill  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES and --pmc GRBM_GUI_ACTIVE (separate passes) of a 2048-id q4 prompt:
#            the matrix pipe's busy share of the prompt GEMM = busy cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
# -> gpurun_out/counters_<tag>/counters.json (+ a text table), copied to profiles/counters.json and profiles/<tag>_counters.txt;
# keyed by the kernel sources' hash like traffic.json, so that bench.py refuses a stale file.
# The program after `--` is python3 itself (no env / bash -c hop: the profiler's preloaded library has initialised the GPU).
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/counters_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( while sleep 40; do echo tick; done ) & HB=$!
LANES="--no-cpu-baseline --no-graph --streams 0 --wide-streams 256 --no-lanes --prefill 0 --generate 0 --serve 0 --steps 4 --warmup 2 --fill prefill"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 420 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/lanes_$C -- python3 $R/bench.py $LANES > $OUT/lanes_$C.json 2> $OUT/lanes_$C.err
  echo "lanes $C rc=$?"
done
PRE="--no-cpu-baseline --no-graph --streams 0 --wide-streams 0 --generate 0 --serve 0 --steps 2 --warmup 1 --fill prefill --prefill 2048"
for C in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/prefill_$C -- python3 $R/bench.py $PRE > $OUT/prefill_$C.json 2> $OUT/prefill_$C.err
  echo "prefill $C rc=$?"
done
kill $HB
python3 $R/tools/counters_from_pmc.py $OUT $TAG
rm -rf $OUT/lanes_FETCH_SIZE $OUT/lanes_WRITE_SIZE $OUT/prefill_SQ_VALU_MFMA_BUSY_CYCLES $OUT/prefill_GRBM_GUI_ACTIVE    # raw counter files: far beyond what gpurun copies back
ls $OUT
