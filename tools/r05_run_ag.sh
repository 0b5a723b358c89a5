#!/bin/bash
# round 5, run AG: every kernel of one headline sort (2^28 pairs) with the gaps between them, no per-kernel events (tools/sort_loop.py)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05ag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -- python3 $R/tools/sort_loop.py --log2 28 --steps 6 --warmup 3 > $OUT/loop.txt 2> $OUT/err.txt
python3 $R/tools/trace_last_sort.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) > $OUT/last_sort_kernels_2p28.txt 2>&1
cp $(find $OUT/prof -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace.csv; rm -rf $OUT/prof
cat $OUT/last_sort_kernels_2p28.txt | cut -c1-140
