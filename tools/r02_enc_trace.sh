cd /tmp && export TMPDIR=/tmp
tag=${1:-enc}
rm -rf $GRAFT_REPO_ROOT/gpurun_out/enc_trace_$tag
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/enc_trace_$tag -- python3 $GRAFT_REPO_ROOT/tools/enc_trunk_time.py ${2:-256} --only-hip 2>&1 | grep "HIP trunk"
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/enc_trace_$tag -name '*kernel_trace.csv' | head -1)
python tools/enc_trace_layers.py $f ${2:-256} ${2:-256} | tee gpurun_out/enc_layers_$tag.txt
rm -rf gpurun_out/enc_trace_$tag
