#!/bin/bash
# round 5, run J: fuzzers on the new paths
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05j
mkdir -p $OUT
timeout 400 python tools/fuzz.py 300 6001 > $OUT/fuzz_library.txt 2>&1
timeout 400 python tools/fuzz_one_object.py 300 6002 > $OUT/fuzz_one_object.txt 2>&1
timeout 300 python tools/fuzz_segments.py 200 6003 > $OUT/fuzz_segments.txt 2>&1
