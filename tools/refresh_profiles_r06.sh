#!/bin/bash
# Round-6 profile set (run on the GPU box through gpurun; everything lands in gpurun_out/r06/; tools/collect_profiles_r06.py copies
# and summarises it into profiles/r06/).  One MI355X per gpurun call.
#   1. bench.py with the driver's flags, then under rocprofv3 --kernel-trace --stats: kernel-stats CSV + the timed region of the trace
#   2. FETCH_SIZE and WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md, HBM section) over bench.py and tools/measure_configs.py
#   3. every kernel of one headline sort, of one C5 sort, of a refused sort (three values) and of a Zipf sort (tools/trace_last_sort.py)
#   4. single-GPU configs, key distributions (32- and 64-bit keys), size ladders, the reference-format ladder
#   5. the in-LDS pass: bucket round against ballot rounds (tools/finish_bucket_bench.hip), all modes
#   6. a rank's compute of the sharded sort (force-dist), key ranges of 1 / 2 / 4 / 8 ranks
set -x
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r06
mkdir -p $OUT
make -C oracle -s > /dev/null 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python bench.py > $OUT/bench_n1_default_flags.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify > $OUT/bench_n1_under_rocprof.json 2> $OUT/prof_bench.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_bench_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2> $OUT/pmc_bf.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_bench_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2> $OUT/pmc_bw.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_cfg_fetch -- python3 $R/tools/measure_configs.py > /dev/null 2> $OUT/pmc_cf.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_cfg_write -- python3 $R/tools/measure_configs.py > /dev/null 2> $OUT/pmc_cw.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -- python3 $R/tools/sort_loop.py --log2 28 --key-bytes 8 --steps 10 --warmup 3 > $OUT/c5_loop_under_rocprof.txt 2> $OUT/prof_c5.err
cd $R
python tools/pmc_summary.py $OUT/pmc_bench_fetch glu_hip > $OUT/pmc_fetch_size_bench.txt
python tools/pmc_summary.py $OUT/pmc_bench_write glu_hip > $OUT/pmc_write_size_bench.txt
python tools/pmc_summary.py $OUT/pmc_cfg_fetch glu_hip > $OUT/pmc_fetch_size_configs.txt
python tools/pmc_summary.py $OUT/pmc_cfg_write glu_hip > $OUT/pmc_write_size_configs.txt
find $OUT/prof_bench -name "*kernel_stats.csv" -exec cp {} $OUT/bench_n1_kernel_stats.csv \;
python tools/trace_summary.py $(find $OUT/prof_bench -name "*kernel_trace.csv" | head -1) --bench 10 > $OUT/bench_n1_timed_region_from_trace.txt
python tools/trace_summary.py $(find $OUT/prof_bench -name "*kernel_trace.csv" | head -1) --sorts 10 >> $OUT/bench_n1_timed_region_from_trace.txt
find $OUT/prof_c5 -name "*kernel_stats.csv" -exec cp {} $OUT/c5_kernel_stats.csv \;
python tools/trace_summary.py $(find $OUT/prof_c5 -name "*kernel_trace.csv" | head -1) --sorts 10 > $OUT/c5_timed_region_from_trace.txt
rm -rf $OUT/prof_bench $OUT/prof_c5 $OUT/pmc_bench_fetch $OUT/pmc_bench_write $OUT/pmc_cfg_fetch $OUT/pmc_cfg_write
tools/r06_trace_one.sh last_sort_kernels_2p28.txt --log2 28
tools/r06_trace_one.sh last_sort_kernels_2p28_u64.txt --log2 28 --key-bytes 8
# (a REFUSED sort, as round 5 refused every input like it: GLU_HIP_SORT_LONG_RUNS=0 takes the segmented passes over long runs away, three
# key values then refuse the attempt on the device and the four ordinary passes run; then the same input with the library's defaults)
GLU_HIP_SORT_LONG_RUNS=0 tools/r06_trace_one.sh refused_sort_kernels_three_values.txt --log2 28 --distinct 3
tools/r06_trace_one.sh three_values_sort_kernels.txt --log2 28 --distinct 3
tools/r06_trace_one.sh zipf_sort_kernels.txt --log2 28 --zipf
tools/r06_trace_one.sh distinct_1000_sort_kernels.txt --log2 28 --distinct-scattered 1000
tools/r06_trace_one.sh distinct_2p20_sort_kernels.txt --log2 28 --distinct-scattered 1048576
tools/r06_trace_one.sh zeros_0p01_sort_kernels.txt --log2 28 --zeros 0.01
tools/r06_trace_one.sh zeros_1_sort_kernels.txt --log2 28 --zeros 1
python tools/measure_configs.py > $OUT/configs_single_gpu.txt 2>&1
python tools/sort_loop.py --log2 28 --key-bytes 8 > $OUT/c5_loop.txt 2>&1
python tools/caller_pairs_probe.py > $OUT/caller_pairs_probe.txt 2>&1
python tools/caller_pairs_probe.py --key-bytes 8 --pairs 8 >> $OUT/caller_pairs_probe.txt 2>&1
python tools/measure_distributions_2p28.py > $OUT/distributions_2p28.txt 2>&1
python tools/measure_distributions_2p28.py 28 u64 > $OUT/distributions_2p28_u64.txt 2>&1
python tools/size_ladder.py pairs 1000 300000000 > $OUT/size_ladder_pairs.txt 2>&1
tests/cpp/bin/bench_ladder > $OUT/bench_ladder_reference_format.txt 2>&1
(cd tools; for a in "4 0" "4 3" "4 2" "4 1" "4 0 12" "8 0" "8 3" "8 2" "8 1" "8 0 40"; do ./finish_bucket_bench 28 $a; done) > $OUT/finish_bucket_bench.txt 2>&1
B="python bench.py --force-dist --log2-keys 27 --no-cpu-baseline --no-one-gpu --pipeline-depth 1"
$B > $OUT/force_dist_world1.json 2> $OUT/fd.err
$B --as-rank-of 2 > $OUT/force_dist_as_rank_of_2.json 2> $OUT/fd.err
$B --as-rank-of 4 > $OUT/force_dist_as_rank_of_4.json 2> $OUT/fd.err
$B --as-rank-of 8 > $OUT/force_dist_as_rank_of_8.json 2> $OUT/fd.err
ls -la $OUT
