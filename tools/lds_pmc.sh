#!/bin/bash
# LDS bank-conflict share of the production scatter kernel (rocprofv3 PMC, harness quick mode)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ldspmc
SB_QUICK=1 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/ldspmc -- $R/tools/scatter_bench 28 > /tmp/ldspmc.log 2>&1
python3 $R/tools/pmc_summary.py /tmp/ldspmc scatter | head -12
