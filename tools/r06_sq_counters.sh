#!/bin/bash
# Round 6: SQ counters of the in-LDS pass's kernel (radix_finish_bucket_kernel), one rocprofv3 --pmc pass per counter pair, over
# tools/sort_loop.py --log2 28 [--key-bytes 8] --steps 3 --warmup 1 -> gpurun_out/r06/finish_bucket_what_bounds_it.txt
# (the same counters as profiles/r05/finish_pass_what_bounds_it.txt has for round 5's ballot-ranked kernel).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RES=$OUT/finish_bucket_what_bounds_it.txt
echo "SQ counters per launch of radix_finish_bucket_kernel (rocprofv3 --pmc, one pass per counter pair, over python tools/sort_loop.py --log2 28" > $RES
echo "[--key-bytes 8] --steps 3 --warmup 1; tools/r06_sq_counters.sh).  One launch orders 2^28 pairs: 65536 workgroups." >> $RES
for kb in 4 8; do
  echo "" >> $RES
  echo "---- $kb-byte keys" >> $RES
  i=0
  for c in "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SALU SQ_INSTS_VALU" "SQ_WAIT_ANY" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES"; do
    i=$((i+1))
    rocprofv3 --pmc $c --output-format csv -d $OUT/sq_$kb_$i -- python3 $R/tools/sort_loop.py --log2 28 --key-bytes $kb --steps 3 --warmup 1 > /dev/null 2> $OUT/sq.err
    python3 $R/tools/pmc_summary.py $OUT/sq_$kb_$i glu_hip | grep -A2 "radix_finish_bucket_kernel<unsigned [a-z]*, [0-9]*, [0-9]*, true, false, false>" | cut -c1-110 >> $RES
    rm -rf $OUT/sq_$kb_$i
  done
done
cat $RES
