#!/bin/bash
# round 5, run O: after the 64-bit threshold moved to 2^23: the sorts' tests, a size ladder of 64-bit keys
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05o
mkdir -p $OUT
python -m pytest tests/test_gpu_lds_finish.py tests/test_gpu_radix_sort.py tests/test_gpu_cpp_api.py -x -q -m gpu > $OUT/t.log 2>&1
python tools/size_ladder.py u64 3000000 300000000 > $OUT/size_ladder_u64.txt 2>&1
timeout 300 python tools/fuzz_one_object.py 200 8001 > $OUT/fuzz_one_object.txt 2>&1
