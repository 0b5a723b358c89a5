#!/bin/bash
# round 5, run AI: every kernel of a refused sort (three key values, 2^28 pairs), with the attempt and without it
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05ai
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -- python3 $R/tools/sort_loop.py --log2 28 --steps 4 --warmup 2 --distinct 3 > $OUT/loop_attempt.txt 2> $OUT/err.txt
python3 $R/tools/trace_last_sort.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) > $OUT/three_values_with_attempt.txt 2>&1
rm -rf $OUT/prof
export GLU_HIP_SORT_LDS_FINISH=0
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -- python3 $R/tools/sort_loop.py --log2 28 --steps 4 --warmup 2 --distinct 3 > $OUT/loop_no_attempt.txt 2> $OUT/err.txt
python3 $R/tools/trace_last_sort.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) > $OUT/three_values_no_attempt.txt 2>&1
rm -rf $OUT/prof
grep -v "dur      [0-9]\.[0-9] " $OUT/three_values_with_attempt.txt | cut -c1-150
echo ----
grep -v "dur      [0-9]\.[0-9] " $OUT/three_values_no_attempt.txt | cut -c1-150
