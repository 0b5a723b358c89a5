#!/bin/bash
# round 5, run N: from which size does the sort that ends in LDS pay at HEAD (it enqueues more launches than in round 4)?
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05n
mkdir -p $OUT
python tools/finish_midsize_probe.py any > $OUT/finish_midsize_any.txt 2>&1
python tools/finish_midsize_probe.py any u64 > $OUT/finish_midsize_any_u64.txt 2>&1
