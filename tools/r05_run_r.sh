#!/bin/bash
# round 5, run R: what one sort of 2^20 / 2^22 / 2^24 pairs launches (kernel trace of the last sort of a loop)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05r
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in 20 22 24; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_$L -- python3 $R/tools/sort_loop.py --log2 $L --steps 10 --warmup 3 > $OUT/loop_$L.txt 2> $OUT/err_$L.txt
  python3 $R/tools/trace_last_sort.py $(find $OUT/prof_$L -name "*kernel_trace.csv" | head -1) > $OUT/last_sort_2p$L.txt 2>&1
  rm -rf $OUT/prof_$L
done
cat $OUT/last_sort_2p20.txt
