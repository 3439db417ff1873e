# rocprofv3 kernel trace of the 64-sequence decode leg (run on the GPU box: gpurun -- bash tools/profile_wide64.sh)
# -> gpurun_out/w64_kernel_stats.csv (copied to profiles/r02_wide64_prefillfill_kernel_stats.csv)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
( while sleep 40; do echo tick; done ) & HB=$!
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_w64 -- python3 $R/bench.py --no-cpu-baseline --streams 0 --wide-streams 64 --prefill 0 --generate 0 --serve 0 --fill prefill --steps 32 --warmup 8 > $R/gpurun_out/w64_bench.json 2> $R/gpurun_out/w64.err
rc=$?
kill $HB
echo rc=$rc
f=$(ls $R/gpurun_out/prof_w64/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/w64_kernel_stats.csv
rm -rf $R/gpurun_out/prof_w64
