# Profiles of the default bench line's kernels, collected on the GPU box:
#     gpurun --timeout 900 -- bash tools/collect_profiles.sh [tag]
# 1. rocprofv3 --kernel-trace --stats of the batch-1 decode step (context filled by ONE prompt-processing call so that the
#    trace holds thousands, not hundreds of thousands, of launches)  -> gpurun_out/prof_<tag>/stats/
# 2. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, SEPARATE passes, eager launches (--no-graph: counters are per dispatch)
#    -> gpurun_out/prof_<tag>/pmc_*/ ; tools/traffic_from_pmc.py turns them into profiles/traffic.json (HBM bytes per launch
#    per kernel family, gfx950 correction applied, keyed by the kernel sources' hash so that bench.py refuses a stale file).
# The program after `--` is python3 itself (no env / bash -c hop: the profiler's preloaded library has initialised the GPU).
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( while sleep 40; do echo tick; done ) & HB=$!
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --brief --fill prefill --steps 64 --warmup 16 > $OUT/bench_stats.json 2> $OUT/bench_stats.err
echo "stats rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py --brief --fill prefill --no-graph --steps 8 --warmup 2 > $OUT/bench_pmc_$C.json 2> $OUT/bench_pmc_$C.err
  echo "pmc $C rc=$?"
done
kill $HB
python3 $R/tools/traffic_from_pmc.py $OUT $TAG
ls $OUT
