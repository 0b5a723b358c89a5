#!/bin/bash
# round 5, run AH: every sort asks (no back-off): tests of the sort that ends in LDS, the reference-format ladder, the distribution sweep
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05ah
mkdir -p $OUT
python -m pytest tests/test_gpu_lds_finish.py tests/test_gpu_radix_sort.py tests/test_gpu_cpp_api.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -n 3 $OUT/pytest.txt
tests/cpp/bin/bench_ladder > $OUT/bench_ladder_reference_format.txt 2>&1
grep "uniform" $OUT/bench_ladder_reference_format.txt | tail -8
python tools/measure_distributions_2p28.py > $OUT/distributions_2p28.txt 2>&1
cat $OUT/distributions_2p28.txt | cut -c1-200
