#!/bin/bash
# Round 6: every kernel of the last sort of tools/sort_loop.py "$@" (rocprofv3 --kernel-trace, tools/trace_last_sort.py) -> gpurun_out/r06/$NAME
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06
NAME=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_one -- python3 $R/tools/sort_loop.py --steps 3 --warmup 2 "$@" > $OUT/$NAME.loop.txt 2> $OUT/prof_one.err
cd $R
python tools/trace_last_sort.py $(find $OUT/prof_one -name "*kernel_trace.csv" | head -1) > $OUT/$NAME
cat $OUT/$NAME.loop.txt >> $OUT/$NAME
rm -rf $OUT/prof_one $OUT/$NAME.loop.txt
