#!/bin/bash
# Round-4 profile set (run on the GPU box through gpurun; summaries are copied from gpurun_out/r04/ into profiles/r04/).
#   1. bench.py under rocprofv3 --kernel-trace --stats (the committed kernel-stats CSV of the bench command)
#   2. FETCH_SIZE and WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md, HBM section) over bench.py (8-bit sort + the
#      4-bit reference-pass-structure sort) and over tools/measure_configs.py (u64 sort, scan, reduce), per kernel
#   3. the same two counters per DISPATCH of the plain count kernel in tools/scatter_bench (SB_R3): once on keys a fill kernel
#      wrote long before, then right behind a scatter that has just written its input with non-temporal stores -- what the
#      FETCH_SIZE of a count kernel says about where its input comes from
#   4. force-dist (the N > 1 path at world size 1) with its own kernel trace
# The oracle / CPU baseline is built beforehand and kept out of the profiled processes (--no-cpu-baseline, --no-verify).
set -x
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r04
mkdir -p $OUT
make -C oracle -s > /dev/null 2>&1
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify > $OUT/bench_n1_bits8_under_rocprof.json 2> $OUT/prof_bench.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_bench_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2> $OUT/pmc_bf.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_bench_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > /dev/null 2> $OUT/pmc_bw.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_cfg_fetch -- python3 $R/tools/measure_configs.py > /dev/null 2> $OUT/pmc_cf.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_cfg_write -- python3 $R/tools/measure_configs.py > /dev/null 2> $OUT/pmc_cw.err
SB_R3=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_sb_fetch -- $R/tools/scatter_bench 28 > /dev/null 2> $OUT/pmc_sb.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dist -- python3 $R/bench.py --force-dist --log2-keys 27 --steps 10 --warmup 3 --pipeline-depth 1 --no-verify --no-transport-fallback > $OUT/bench_force_dist_2p27_under_rocprof.json 2> $OUT/prof_dist.err
cd $R
python tools/pmc_summary.py $OUT/pmc_bench_fetch glu_hip > $OUT/pmc_fetch_size_bench.txt
python tools/pmc_summary.py $OUT/pmc_bench_write glu_hip > $OUT/pmc_write_size_bench.txt
python tools/pmc_summary.py $OUT/pmc_cfg_fetch glu_hip > $OUT/pmc_fetch_size_configs.txt
python tools/pmc_summary.py $OUT/pmc_cfg_write glu_hip > $OUT/pmc_write_size_configs.txt
python - > $OUT/pmc_fetch_size_count_kernel_per_dispatch.txt <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/pmc_sb_fetch/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "radix_count_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
print("FETCH_SIZE (KB) of every radix_count_kernel<u32, 8 or 4, 1024 threads> dispatch of SB_R3=1 tools/scatter_bench 28, in dispatch order")
print("(2^28 keys = 1 GiB = 1048576 KB.  In each run_lines: dispatch 1 counts the keys the fill kernel wrote at start-up, the following")
print(" five count shift + bits of what the scatter launched just before them has written with non-temporal stores)")
for r in rows:
    print("dispatch %6s  %-40s FETCH_SIZE %12.0f KB" % (r["Dispatch_Id"], r["Kernel_Name"][:40], float(r["Counter_Value"])))
PY
python tools/measure_configs.py > $OUT/configs_single_gpu.txt 2>&1
python bench.py --force-dist --log2-keys 27 --no-cpu-baseline > $OUT/bench_force_dist_2p27.json 2> $OUT/bench_force_dist.err
find $OUT/prof_bench -name "*kernel_stats.csv" -exec cp {} $OUT/bench_n1_bits8_kernel_stats.csv \;
# the stats above average over the WHOLE process (the calibration sorts of prepare, the four-pass and the 4-bit comparison legs);
# the timed region of the bench is the last 10 sorts that ended in LDS: their launches from the kernel trace of the same run
python tools/trace_summary.py $(find $OUT/prof_bench -name "*kernel_trace.csv" | head -1) --bench 10 > $OUT/bench_n1_bits8_timed_region_from_trace.txt
find $OUT/prof_dist -name "*kernel_stats.csv" -exec cp {} $OUT/force_dist_2p27_kernel_stats.csv \;
rm -rf $OUT/prof_bench $OUT/prof_dist $OUT/pmc_bench_fetch $OUT/pmc_bench_write $OUT/pmc_cfg_fetch $OUT/pmc_cfg_write $OUT/pmc_sb_fetch
ls -la $OUT
