# rocprofv3 kernel stats of ONE prompt of P ids through the host model (run on the GPU box:
#     gpurun -- bash tools/profile_prefill.sh 256 ) -> gpurun_out/pf<P>_kernel_stats.csv
P=${1:-256}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pf$P -- python3 $R/tools/prefill_one.py $P 10 > $R/gpurun_out/pf$P.out 2> $R/gpurun_out/pf$P.err || exit 1
cat $R/gpurun_out/pf$P.out
f=$(ls $R/gpurun_out/prof_pf$P/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/pf${P}_kernel_stats.csv
rm -rf $R/gpurun_out/prof_pf$P
