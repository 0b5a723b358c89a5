#!/bin/bash
# round 5, run U: plain against non-temporal line stores over mid sizes for keys-only and 64-bit-key sorts; pairs at the new defaults
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05u
mkdir -p $OUT
for M in keys u64; do
  GLU_HIP_SORT_NT_MIN_BYTES=0 python tools/size_ladder.py $M 3000000 200000000 > $OUT/ladder_${M}_nt.txt 2>&1
  GLU_HIP_SORT_NT_MIN_BYTES=1000000000000 python tools/size_ladder.py $M 3000000 200000000 > $OUT/ladder_${M}_plain.txt 2>&1
done
python tools/size_ladder.py pairs 1000000 80000000 > $OUT/ladder_pairs_default.txt 2>&1
python tools/geometry_switch_ladder.py 3900000 4400000 1.01 > $OUT/switch_default.txt 2>&1
python -m pytest tests/test_gpu_radix_sort.py -x -q -m gpu > $OUT/pytest_sort.txt 2>&1
tail -3 $OUT/pytest_sort.txt
