#!/bin/bash
# PMC passes (separate, kernel-trace only) for one GEMM shape: tools/pmc_round.sh <tag>   (env PMC_SHAPE, PMC_BF16_OUT, PMC_KERNEL)
tag=$1
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${tag}_f -- python $R/tools/pmc_gemm.py > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${tag}_w -- python $R/tools/pmc_gemm.py > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/${tag}_f $O/${tag}_w $O/${tag}_pmc.json | tail -12
rm -rf $O/${tag}_f $O/${tag}_w
