#!/bin/bash
# round 5, run AJ: the count kernels' peel mode with memory: few-distinct-keys tests, then three / two / seven / twenty key values at 2^28 (kernel trace), uniform for comparison
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05aj
mkdir -p $OUT
python -m pytest tests/test_gpu_radix_sort.py tests/test_gpu_segmented_sort.py -x -q -m gpu -k "distinct or distribution or few or duplicate or segment or paired or reference" > $OUT/pytest.txt 2>&1
tail -n 2 $OUT/pytest.txt
cd /tmp && export TMPDIR=/tmp
for D in 3 7 12 20 64; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -- python3 $R/tools/sort_loop.py --log2 28 --steps 4 --warmup 2 --distinct $D > $OUT/loop_$D.txt 2> $OUT/err.txt
  python3 $R/tools/trace_last_sort.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) > $OUT/distinct_$D.txt 2>&1
  rm -rf $OUT/prof
  echo "distinct $D: $(tail -n 1 $OUT/loop_$D.txt | cut -c1-110)"
  grep "count_kernel" $OUT/distinct_$D.txt | grep -v "dur      [0-9]\.[0-9] " | cut -c1-120
done
