set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_r01c.log 2>&1; tail -2 gpurun_out/pytest_gpu_r01c.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_r01c.log 2>&1; tail -1 gpurun_out/smoke_r01c.log
python bench.py > gpurun_out/bench_r01c.json 2> gpurun_out/bench_r01c.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01c -- python3 $R/bench.py --steps 10 --warmup 3 > $R/gpurun_out/prof_bench_c.json 2> $R/gpurun_out/prof_bench_c.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_r01c_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-alt > /dev/null 2> $R/gpurun_out/pmc_fc.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_r01c_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-alt > /dev/null 2> $R/gpurun_out/pmc_wc.err
cd $R
python tools/pmc_summary.py gpurun_out/pmc_r01c_fetch > gpurun_out/pmc_fetch_r01c.txt
python tools/pmc_summary.py gpurun_out/pmc_r01c_write > gpurun_out/pmc_write_r01c.txt
python tools/measure_configs.py > gpurun_out/configs_r01c.txt 2>&1
find gpurun_out/prof_r01c -name "*kernel_stats.csv" | head -2
