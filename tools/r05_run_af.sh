#!/bin/bash
# round 5, run AF: kernel sequence of a keys-only sort of 0.8 M and 1.2 M keys (a step in the size ladder between them)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05af
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in 20.2; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_$L -- python3 $R/tools/sort_loop.py --log2 $L --steps 10 --warmup 3 --keys-only > $OUT/loop_$L.txt 2> $OUT/err_$L.txt
  python3 $R/tools/trace_last_sort.py $(find $OUT/prof_$L -name "*kernel_trace.csv" | head -1) > $OUT/last_sort_keys_2p$L.txt 2>&1
  rm -rf $OUT/prof_$L
  cat $OUT/last_sort_keys_2p$L.txt | cut -c1-150
done
