cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/u64diag
mkdir -p $O
for nt in 1 0; do
  export GLU_HIP_SORT_NT_STORES=$nt
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f$nt -- python3 $R/tools/u64_probe.py 28 > $O/probe_f$nt.txt 2> $O/err_f$nt.txt
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w$nt -- python3 $R/tools/u64_probe.py 28 > $O/probe_w$nt.txt 2> $O/err_w$nt.txt
  python3 $R/tools/pmc_each.py $O/f$nt "scatter_lines_kernel<unsigned long" > $O/each_f$nt.txt
  python3 $R/tools/pmc_each.py $O/w$nt "scatter_lines_kernel<unsigned long" > $O/each_w$nt.txt
  rm -rf $O/f$nt $O/w$nt
done
unset GLU_HIP_SORT_NT_STORES
python3 $R/tools/u64_probe.py 28 > $O/probe_plain.txt 2>&1
