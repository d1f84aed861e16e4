cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/e2eprof; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/e2e_profile.py 256 4 10 > $O/run.log 2>&1
cd $R
python tools/summarize_rocprof.py $O gpurun_out/e2e_256_kernel_stats.csv "tools/e2e_profile.py 256 4 10 (whole DIINN forward, 10 calls)" 2>&1 | tail -2
head -14 gpurun_out/e2e_256_kernel_stats.csv | cut -c1-150
