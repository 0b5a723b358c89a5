#!/bin/bash
# Round-2 profile set (run on the GPU box through gpurun; summaries are copied from gpurun_out/ into profiles/r02/).
#   1. bench.py under rocprofv3 --kernel-trace --stats (the committed kernel-stats CSV of the bench command)
#   2. FETCH_SIZE and WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md, HBM section) over bench.py (8-bit sort +
#      the 4-bit reference-pass-structure sort) and over tools/measure_configs.py (u64 sort, keys of every width, scan,
#      reduce), summarised per kernel by tools/pmc_summary.py
# The oracle / CPU baseline is built beforehand and kept out of the profiled processes (--no-cpu-baseline, --no-verify).
set -x
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r02
mkdir -p $OUT
make -C oracle -s > /dev/null 2>&1
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify > $OUT/bench_n1_bits8_under_rocprof.json 2> $OUT/prof_bench.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_bench_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2> $OUT/pmc_bf.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_bench_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2> $OUT/pmc_bw.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_cfg_fetch -- python3 $R/tools/measure_configs.py > /dev/null 2> $OUT/pmc_cf.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_cfg_write -- python3 $R/tools/measure_configs.py > /dev/null 2> $OUT/pmc_cw.err
cd $R
python tools/pmc_summary.py $OUT/pmc_bench_fetch glu_hip > $OUT/pmc_fetch_size_bench.txt
python tools/pmc_summary.py $OUT/pmc_bench_write glu_hip > $OUT/pmc_write_size_bench.txt
python tools/pmc_summary.py $OUT/pmc_cfg_fetch glu_hip > $OUT/pmc_fetch_size_configs.txt
python tools/pmc_summary.py $OUT/pmc_cfg_write glu_hip > $OUT/pmc_write_size_configs.txt
python tools/measure_configs.py > $OUT/configs_single_gpu.txt 2>&1
python bench.py --force-dist --log2-keys 27 --no-cpu-baseline > $OUT/bench_force_dist_2p27.json 2> $OUT/bench_force_dist.err
find $OUT/prof_bench -name "*kernel_stats.csv" -exec cp {} $OUT/bench_n1_bits8_kernel_stats.csv \;
rm -rf $OUT/prof_bench $OUT/pmc_bench_fetch $OUT/pmc_bench_write $OUT/pmc_cfg_fetch $OUT/pmc_cfg_write
ls -la $OUT
