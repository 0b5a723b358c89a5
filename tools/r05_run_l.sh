#!/bin/bash
# round 5, run L: long fuzz at HEAD (one long-lived object: long runs, zero shares, 64-bit ties, small ranges; segmented sort; library surface)
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05l
mkdir -p $OUT
timeout 1000 python tools/fuzz_one_object.py 900 7001 > $OUT/fuzz_one_object_long.txt 2>&1
timeout 500 python tools/fuzz_segments.py 400 7002 > $OUT/fuzz_segments_long.txt 2>&1
timeout 500 python tools/fuzz.py 400 7003 > $OUT/fuzz_library_long.txt 2>&1
