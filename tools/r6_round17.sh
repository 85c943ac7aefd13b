#!/bin/bash
# sparse-table RoIPool: parity test, then the timing table
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "sparse_table or roi_pool" 2>&1 | tail -15 > $O/r6_28_tests.txt
cat $O/r6_28_tests.txt
timeout 600 python tools/roi_st_bench.py 2>&1 | grep -v amdgpu.ids > $O/r6_28_roi_st.txt
cat $O/r6_28_roi_st.txt
