#!/bin/bash
# Round-5 profile set (run on the GPU box through gpurun; everything lands in gpurun_out/r05/ and is copied into profiles/r05/).
#   1. bench.py (the driver's command), then the same under rocprofv3 --kernel-trace --stats: kernel-stats CSV + the timed region
#      of the trace (tools/trace_summary.py)
#   2. FETCH_SIZE and WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md, HBM section) over bench.py and over
#      tools/measure_configs.py (u64 sort, scan, reduce), per kernel -> tools/make_traffic_json.py r05
#   3. C5 (2^28 u64 + u32) under rocprofv3 --kernel-trace: per-kernel durations of the timed sorts
#   4. the N > 1 path at world size 1 (force-dist): a rank's compute with the key range of world 1 / 2 / 4 / 8, and the variants for
#      the eight-rank shape
#   5. key distributions at 2^28, size ladders, events A/B of the bench, phase clocks of the in-LDS pass, fuzzers
set -x
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05
mkdir -p $OUT
make -C oracle -s > /dev/null 2>&1
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python bench.py --no-cpu-baseline --no-alt --full-kernel-events > $OUT/bench_n1_events_everywhere.json 2> /dev/null
python bench.py --no-cpu-baseline --no-alt > $OUT/bench_n1_events_light.json 2> /dev/null
python bench.py --no-cpu-baseline --no-alt --no-kernel-events > $OUT/bench_n1_no_events.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify > $OUT/bench_n1_under_rocprof.json 2> $OUT/prof_bench.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_bench_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2> $OUT/pmc_bf.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_bench_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2> $OUT/pmc_bw.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_cfg_fetch -- python3 $R/tools/measure_configs.py > /dev/null 2> $OUT/pmc_cf.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_cfg_write -- python3 $R/tools/measure_configs.py > /dev/null 2> $OUT/pmc_cw.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -- python3 $R/tools/sort_loop.py --log2 28 --key-bytes 8 --steps 10 --warmup 3 > $OUT/c5_loop_under_rocprof.txt 2> $OUT/prof_c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dist -- python3 $R/bench.py --force-dist --log2-keys 27 --steps 10 --warmup 3 --pipeline-depth 1 --no-verify --no-transport-fallback --no-one-gpu --no-cpu-baseline > $OUT/bench_force_dist_2p27_under_rocprof.json 2> $OUT/prof_dist.err
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
find $OUT/prof_dist -name "*kernel_stats.csv" -exec cp {} $OUT/force_dist_2p27_kernel_stats.csv \;
rm -rf $OUT/prof_bench $OUT/prof_c5 $OUT/prof_dist $OUT/pmc_bench_fetch $OUT/pmc_bench_write $OUT/pmc_cfg_fetch $OUT/pmc_cfg_write
python tools/measure_configs.py > $OUT/configs_single_gpu.txt 2>&1
python tools/sort_loop.py --log2 28 --key-bytes 8 > $OUT/c5_loop.txt 2>&1
B="python bench.py --force-dist --log2-keys 27 --no-cpu-baseline --no-one-gpu --pipeline-depth 1"
$B > $OUT/force_dist_world1.json 2> $OUT/fd.err
GLU_HIP_SEG_LDS_FINISH=0 $B > $OUT/force_dist_world1_three_passes.json 2> $OUT/fd.err
$B --as-rank-of 2 > $OUT/force_dist_as_rank_of_2.json 2> $OUT/fd.err
$B --as-rank-of 4 > $OUT/force_dist_as_rank_of_4.json 2> $OUT/fd.err
$B --as-rank-of 8 > $OUT/force_dist_as_rank_of_8.json 2> $OUT/fd.err
GLU_HIP_SEG_MAX_GEO=5 $B --as-rank-of 8 > $OUT/force_dist_as_rank_of_8_tile17408.json 2> $OUT/fd.err
GLU_HIP_SEG_SPLIT_MAX=3 GLU_HIP_SEG_SPLIT_GEO=4 $B --as-rank-of 8 > $OUT/force_dist_as_rank_of_8_split2.json 2> $OUT/fd.err
GLU_HIP_SEG_SPLIT_MAX=3 $B --as-rank-of 8 > $OUT/force_dist_as_rank_of_8_split4.json 2> $OUT/fd.err
python bench.py --force-dist --log2-keys 27 --no-cpu-baseline > $OUT/bench_force_dist_2p27.json 2> $OUT/fd.err
python tools/measure_distributions_2p28.py > $OUT/distributions_2p28.txt 2>&1
python tools/size_ladder.py pairs 1000 300000000 > $OUT/size_ladder_pairs.txt 2>&1
tests/cpp/bin/bench_ladder > $OUT/bench_ladder_reference_format.txt 2>&1
tools/finish_stamps_bench 28 4 > $OUT/finish_stamps_u32.txt 2>&1
tools/finish_stamps_bench 28 8 16 > $OUT/finish_stamps_u64_rank16.txt 2>&1
tools/finish_stamps_bench 28 8 48 > $OUT/finish_stamps_u64_all_rounds.txt 2>&1
timeout 200 python tools/fuzz.py 150 5005 > $OUT/fuzz_library.txt 2>&1
timeout 200 python tools/fuzz_one_object.py 150 5006 > $OUT/fuzz_one_object.txt 2>&1
timeout 150 python tools/fuzz_segments.py 100 5007 > $OUT/fuzz_segments.txt 2>&1
python -m pytest tests -q -m gpu > $OUT/pytest_gpu_head.txt 2>&1
ls -la $OUT
