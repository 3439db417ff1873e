# rocprofv3 kernel trace of the 256-sequence decode leg (two lanes of 128 for q8 / q4): gpurun -- bash tools/profile_lanes.sh [streams]
# -> gpurun_out/lanes<streams>_kernel_stats.csv (copied to profiles/r03b_lanes256_kernel_stats.csv)
W=${1:-256}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
( while sleep 40; do echo tick; done ) & HB=$!
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lanes -- python3 $R/bench.py --no-cpu-baseline --streams 0 --wide-streams $W --no-lanes --prefill 0 --generate 0 --serve 0 --steps 32 --warmup 8 > $R/gpurun_out/lanes_bench.json 2> $R/gpurun_out/lanes.err
rc=$?
kill $HB
echo rc=$rc
f=$(ls $R/gpurun_out/prof_lanes/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/lanes${W}_kernel_stats.csv
rm -rf $R/gpurun_out/prof_lanes
