#!/bin/bash
# round 5, run AL: distribution sweep and few-distinct-values timings after the quicker give-up of the count kernels' peel mode
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05al
mkdir -p $OUT
python -m pytest tests/test_gpu_radix_sort.py tests/test_gpu_segmented_sort.py tests/test_gpu_lds_finish.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -n 1 $OUT/pytest.txt
python tools/measure_distributions_2p28.py > $OUT/distributions_2p28.txt 2>&1
cut -c1-120 $OUT/distributions_2p28.txt | tail -14
for D in 2 3 7 8 9 12 20 64 1000; do echo "distinct $D: $(python tools/sort_loop.py --log2 28 --steps 5 --warmup 2 --distinct $D | tail -n 1 | cut -c60-130)"; done > $OUT/distinct_values_2p28.txt 2>&1
cat $OUT/distinct_values_2p28.txt
