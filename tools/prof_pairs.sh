cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pairs -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify --no-alt > $R/gpurun_out/prof_pairs.json 2> $R/gpurun_out/prof_pairs.err
find $R/gpurun_out/prof_pairs -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof_pairs_stats.csv \;
rm -rf $R/gpurun_out/prof_pairs
cut -c1-90,300-420 $R/gpurun_out/prof_pairs_stats.csv | head -12
