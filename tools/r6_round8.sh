#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_real_size_gpu.py -m gpu -q -x > $O/r6_08_real_size.log 2>&1; tail -15 $O/r6_08_real_size.log
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "pp8 or linear_act or bias_act" > $O/r6_08_tests.log 2>&1; tail -3 $O/r6_08_tests.log
PROF=1 VARIANTS=1 timeout 300 python tools/pp8_probe.py fc7 2>&1 | grep -v amdgpu.ids | grep -v "wave [1-35-7]" > $O/r6_08_pp8_probe.txt; cat $O/r6_08_pp8_probe.txt
