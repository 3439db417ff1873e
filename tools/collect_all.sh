# Every profile behind the bench line's roofline objects, on the CURRENT kernel sources (run on the GPU box, after the last source change:
#     gpurun --timeout 1200 -- bash tools/collect_all.sh r05 ):
#   1. batch-1 decode step, per configuration (q4 q8 f16): rocprofv3 --kernel-trace --stats -> <tag>_fused_<mode>_kernel_stats.csv, and
#      --pmc FETCH_SIZE / --pmc WRITE_SIZE in SEPARATE passes (eager launches: counters are per dispatch) -> traffic.json
#      (HBM bytes per launch per kernel family, gfx950 correction applied: tools/traffic_from_pmc.py)
#   2. the multi-sequence legs (8, 64, 256 sequences; q4 -- a 512-sequence FETCH_SIZE pass ended at its time limit once in round 5: left out): the same two counters over the decoder's dispatches -> counters.json
#      sections lanes8 / lanes64 / lanes256 (HBM bytes per step: tools/counters_from_pmc.py), and the kernel trace of the
#      256-sequence step -> <tag>_lanes256_kernel_stats.csv
#   3. prompts: kernel traces of a 512- and a 2048-id prompt -> <tag>_prefill{512,2048}_kernel_stats.csv; SQ_VALU_MFMA_BUSY_CYCLES and
#      GRBM_GUI_ACTIVE (separate passes) of the 2048-id prompt -> counters.json section prefill2048
#   4. the 256-sequence step's memory-side read requests BY SIZE (TCC_EA0_RDREQ_32B / _64B / _128B, one pass each): what the guide's
#      "double FETCH_SIZE" is calibrated against for kernels whose reads are not all 16 B per lane -> section lanes256_by_request_size
# Both json files are keyed by bench.csrc_fingerprint(): bench.py refuses them after any change of kernel code.
# The program after `--` is python3 itself (no env / bash -c hop: the profiler's preloaded library has initialised the GPU);
# --pmc runs carry --kernel-trace only.  Raw counter files are deleted here: far beyond what gpurun copies back.
TAG=${1:-r05}
MODES=${2:-"q4 q8 f16"}
PARTS=${3:-"1 2 3 4"}          # which of the groups below (a group re-collected alone: "2 4" etc.; the others keep their sections)
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
[ -f $OUT/counters.json ] || cp $R/profiles/counters.json $OUT/ 2>/dev/null     # (sections of groups not re-collected now: kept when the fingerprint agrees)
cd /tmp && export TMPDIR=/tmp
( while sleep 40; do echo tick; done ) & HB=$!
trap "kill $HB" EXIT
RP="rocprofv3 --output-format csv"
first_csv() { ls $1/*/*kernel_stats.csv 2>/dev/null | head -1; }

# ---- 1. the batch-1 step
if has 1; then
for M in $MODES; do
  timeout -k 10 300 $RP --kernel-trace --stats -d $OUT/stats_$M -- python3 $R/bench.py --brief --mode $M --fill prefill --steps 64 --warmup 16 > $OUT/bench_stats_$M.json 2> $OUT/bench_stats_$M.err
  echo "stats $M rc=$?"
  cp "$(first_csv $OUT/stats_$M)" $OUT/${TAG}_fused_${M}_kernel_stats.csv; rm -rf $OUT/stats_$M
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 $RP --kernel-trace --pmc $C -d $OUT/pmc_${M}_$C -- python3 $R/bench.py --brief --mode $M --fill prefill --no-graph --steps 8 --warmup 2 > $OUT/bench_pmc_${M}_$C.json 2> $OUT/bench_pmc_${M}_$C.err
    echo "pmc $M $C rc=$?"
  done
done
python3 $R/tools/traffic_from_pmc.py $OUT $TAG $MODES || exit 1
rm -rf $OUT/pmc_*
fi

# ---- 2. the multi-sequence legs
if has 2; then
COMMON="--no-cpu-baseline --no-graph --no-lanes --prefill 0 --generate 0 --serve 0 --steps 4 --warmup 2 --fill prefill"
for S in 8 64 256; do
  if [ $S -eq 8 ]; then SEL="--streams 8 --wide-streams 0"; else SEL="--streams 0 --wide-streams $S"; fi
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 420 $RP --kernel-trace --pmc $C -d $OUT/lanes${S}_$C -- python3 $R/bench.py $COMMON $SEL > $OUT/lanes${S}_$C.json 2> $OUT/lanes${S}_$C.err
    echo "lanes$S $C rc=$?"
  done
done
timeout -k 10 400 $RP --kernel-trace --stats -d $OUT/stats_lanes256 -- python3 $R/bench.py --no-cpu-baseline --streams 0 --wide-streams 256 --no-lanes --prefill 0 --generate 0 --serve 0 --steps 32 --warmup 8 --fill prefill > $OUT/lanes256_bench.json 2> $OUT/lanes256_bench.err
echo "stats lanes256 rc=$?"
cp "$(first_csv $OUT/stats_lanes256)" $OUT/${TAG}_lanes256_kernel_stats.csv; rm -rf $OUT/stats_lanes256
fi

# ---- 3. prompts
if has 3; then
for P in 512 2048; do
  timeout -k 10 200 $RP --kernel-trace --stats -d $OUT/stats_pf$P -- python3 $R/tools/prefill_one.py $P 10 > $OUT/pf$P.out 2> $OUT/pf$P.err
  echo "stats prefill $P rc=$?"
  cp "$(first_csv $OUT/stats_pf$P)" $OUT/${TAG}_prefill${P}_kernel_stats.csv; rm -rf $OUT/stats_pf$P
done
PRE="--no-cpu-baseline --no-graph --streams 0 --wide-streams 0 --generate 0 --serve 0 --steps 2 --warmup 1 --fill prefill --prefill 2048"
for C in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  timeout -k 10 300 $RP --kernel-trace --pmc $C -d $OUT/prefill_$C -- python3 $R/bench.py $PRE > $OUT/prefill_$C.json 2> $OUT/prefill_$C.err
  echo "prefill $C rc=$?"
done
fi
# ---- 4. read requests by size
if has 4; then
COMMON="--no-cpu-baseline --no-graph --no-lanes --prefill 0 --generate 0 --serve 0 --steps 4 --warmup 2 --fill prefill"
for C in TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum; do
  timeout -k 10 420 $RP --kernel-trace --pmc $C -d $OUT/lanes256_$C -- python3 $R/bench.py $COMMON --streams 0 --wide-streams 256 > $OUT/lanes256_$C.json 2> $OUT/lanes256_$C.err
  echo "lanes256 $C rc=$?"
done
fi
python3 $R/tools/counters_from_pmc.py $OUT $TAG || exit 1
rm -rf $OUT/lanes256_TCC_EA0_RDREQ_* $OUT/lanes*_FETCH_SIZE $OUT/lanes*_WRITE_SIZE $OUT/prefill_SQ_VALU_MFMA_BUSY_CYCLES $OUT/prefill_GRBM_GUI_ACTIVE
ls -la $OUT
