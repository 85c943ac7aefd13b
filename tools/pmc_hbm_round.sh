#!/bin/bash
# PMC passes (separate, kernel-trace only) for the HBM-bound kernels: tools/pmc_hbm_round.sh <tag>
tag=$1
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${tag}_f -- python $R/tools/pmc_hbm.py > $O/${tag}_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${tag}_w -- python $R/tools/pmc_hbm.py > $O/${tag}_w.log 2>&1
cd $R
python tools/pmc_hbm_summary.py $O/${tag}_f $O/${tag}_w $O/${tag}_pmc_hbm.json | tail -40
rm -rf $O/${tag}_f $O/${tag}_w
