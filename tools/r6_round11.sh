#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
bash tools/ab_bench.sh r6_11_fc7 "--no-side --steps 200" "--no-side --steps 200 --engine-opt fused_fc7_fwd=0" 3
timeout 600 python tools/roi_align_bench.py 2>&1 | grep -v amdgpu.ids > $O/r6_11_roi_align.txt; cat $O/r6_11_roi_align.txt
timeout 900 python bench.py > $O/r6_11_bench.json 2> $O/r6_11_bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r6_11_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k in d:
    if k.startswith('side_'): print(k, d[k] if not isinstance(d[k], dict) else {kk: d[k][kk] for kk in d[k] if kk not in ('how','what')})
PY
