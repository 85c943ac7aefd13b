#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
get() { python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%.1f img/s  %.4f ms' % (d['value'], d['ms_per_step']))"; }
: > $O/r6_14_nwg.txt
for i in 1 2; do
  for t in "" "--tune 18=248" "--tune 18=240" "--tune 18=224"; do
    echo "[$t]: $(python bench.py --no-cpu-baseline --no-side --steps 200 $t 2>/dev/null | get)" | tee -a $O/r6_14_nwg.txt
  done
done
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/nwg_prof -- python $R/bench.py --steps 40 --no-cpu-baseline --no-side --tune 18=240 > /dev/null 2>&1)
python tools/prof_timeline.py /tmp/nwg_prof 1 2>&1 | grep -v "conv_\|maxpool\|preprocess" | tail -22 > $O/r6_14_nwg_timeline.txt; cat $O/r6_14_nwg_timeline.txt
