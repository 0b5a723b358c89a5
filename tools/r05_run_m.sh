#!/bin/bash
# round 5, run M: the driver's round-end sequence at HEAD: smoke, the whole GPU suite, the bench
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05m
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
