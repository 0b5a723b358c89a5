#!/bin/bash
# round 5, run I: key-distribution sweep at 2^28, SQ counters of the in-LDS pass (u32 and u64), LDS finish tests
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05i
mkdir -p $OUT
python -m pytest tests/test_gpu_lds_finish.py -x -q -m gpu > $OUT/t_fin.log 2>&1
python tools/measure_distributions_2p28.py > $OUT/distributions_2p28.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_u32_$tag -- python3 $R/tools/sort_loop.py --log2 28 --steps 3 --warmup 1 > /dev/null 2> $OUT/pmc_u32_$tag.err
  python3 $R/tools/pmc_summary.py $OUT/pmc_u32_$tag radix_finish_sort > $OUT/pmc_finish_u32_$tag.txt 2>&1
  python3 $R/tools/pmc_summary.py $OUT/pmc_u32_$tag "radix_scatter_lines_kernel<unsigned int, 8" > $OUT/pmc_scatter_u32_$tag.txt 2>&1
  rm -rf $OUT/pmc_u32_$tag
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_u64_$tag -- python3 $R/tools/sort_loop.py --log2 28 --key-bytes 8 --steps 3 --warmup 1 > /dev/null 2> $OUT/pmc_u64_$tag.err
  python3 $R/tools/pmc_summary.py $OUT/pmc_u64_$tag radix_finish_sort > $OUT/pmc_finish_u64_$tag.txt 2>&1
  rm -rf $OUT/pmc_u64_$tag
done
