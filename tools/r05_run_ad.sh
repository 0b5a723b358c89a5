#!/bin/bash
# round 5, run AD: fuzzers at HEAD after the mid-size retuning (sizes around the moved switch points; limit 2^23 and, FUZZ_LARGE, most sizes in [2^22, 2^23])
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05ad
mkdir -p $OUT
timeout 280 python tools/fuzz.py 240 9201 > $OUT/fuzz_library_retuned.txt 2>&1
FUZZ_LARGE=1 timeout 280 python tools/fuzz.py 240 9202 > $OUT/fuzz_library_large_retuned.txt 2>&1
timeout 280 python tools/fuzz_one_object.py 240 9203 > $OUT/fuzz_one_object_retuned.txt 2>&1
for f in $OUT/*.txt; do tail -n 1 $f; done
