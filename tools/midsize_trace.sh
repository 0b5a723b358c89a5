cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/mtrace
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/mtrace -- python3 $R/tools/midsize_trace.py ${1:-20} 2> /dev/null
f=$(find $R/gpurun_out/mtrace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/midsize_trace.py --analyze $f ${2:-12}
rm -rf $R/gpurun_out/mtrace
