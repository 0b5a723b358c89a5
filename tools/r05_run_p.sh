#!/bin/bash
# round 5, run P: long fuzz at HEAD with seeds no earlier run used
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05p
mkdir -p $OUT
timeout 460 python tools/fuzz.py 420 9101 > $OUT/fuzz_library_long.txt 2>&1
timeout 460 python tools/fuzz_one_object.py 420 9102 > $OUT/fuzz_one_object_long.txt 2>&1
timeout 340 python tools/fuzz_segments.py 300 9103 > $OUT/fuzz_segments_long.txt 2>&1
for f in $OUT/*.txt; do tail -n 2 $f; done
