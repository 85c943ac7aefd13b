#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --hip-trace -d /tmp/gap_prof -- python $R/bench.py --steps 40 --no-cpu-baseline --no-side > /dev/null 2> $O/r6_12_gap.err)
python tools/gap_probe.py /tmp/gap_prof schema > $O/r6_12_gap_schema.txt 2>&1
python tools/gap_probe.py /tmp/gap_prof > $O/r6_12_gap.txt 2>&1
head -c 3000 $O/r6_12_gap_schema.txt
bash tools/pmc_kernel.sh $O/r6_12_pmc_conv_dc5.txt "%kernel%" python $R/tools/conv_trace_dc5.py > /dev/null 2>&1
bash tools/pmc_kernel.sh $O/r6_12_pmc_fc7.txt "%pp8%" python $R/tools/linear_bench.py > /dev/null 2>&1
tail -3 $O/r6_12_pmc_fc7.txt | cut -c1-400
