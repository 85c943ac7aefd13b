#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
get() { python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%.1f img/s  %.4f ms' % (d['value'], d['ms_per_step']))"; }
for i in 1 2 3; do
  echo "plain      : $(python bench.py --no-cpu-baseline --no-side --steps 200 2>/dev/null | get)" | tee -a $O/r6_13_tail_fill.txt
  echo "fill 200 MB: $(DRN_PROBE_TAIL_FILL=200 python bench.py --no-cpu-baseline --no-side --steps 200 2>/dev/null | get)" | tee -a $O/r6_13_tail_fill.txt
done
export TMPDIR=/tmp
(cd /tmp && DRN_PROBE_TAIL_FILL=200 timeout 600 rocprofv3 --kernel-trace -d /tmp/fill_prof -- python $R/bench.py --steps 40 --no-cpu-baseline --no-side > /dev/null 2>&1)
python tools/prof_timeline.py /tmp/fill_prof 1 2>&1 | grep -v "conv_\|maxpool\|preprocess" | tail -40 > $O/r6_13_tail_fill_timeline.txt; cat $O/r6_13_tail_fill_timeline.txt
