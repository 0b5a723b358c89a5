#!/bin/bash
# Round 6: FETCH_SIZE / WRITE_SIZE per kernel of the headline sort and of C5, separate --pmc passes (MI355X_MICROARCH.md, HBM section).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_u32_$c -- python3 $R/tools/sort_loop.py --log2 28 --steps 3 --warmup 1 > /dev/null 2> $OUT/pmc_u32_$c.err
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_u64_$c -- python3 $R/tools/sort_loop.py --log2 28 --key-bytes 8 --steps 3 --warmup 1 > /dev/null 2> $OUT/pmc_u64_$c.err
done
cd $R
for t in u32 u64; do
  python tools/pmc_summary.py $OUT/pmc_${t}_FETCH_SIZE glu_hip > $OUT/pmc_fetch_size_$t.txt
  python tools/pmc_summary.py $OUT/pmc_${t}_WRITE_SIZE glu_hip > $OUT/pmc_write_size_$t.txt
done
rm -rf $OUT/pmc_u32_* $OUT/pmc_u64_*
grep -A2 'radix_finish_bucket_kernel\|radix_scatter_lines_kernel<unsigned [a-z]*, 8, 1024, 1[02], false, true, 0, false, 4, true, true, 0, false, false, false' $OUT/pmc_fetch_size_*.txt $OUT/pmc_write_size_*.txt | cut -c1-230
