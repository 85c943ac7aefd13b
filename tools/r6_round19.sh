#!/bin/bash
# sparse-table RoIPool on the DC5 map: kernel split (prep / chunk-major copy / pooling) and PMC attribution
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp ROI_C=2048 ROI_STRIDE=8
for r in 250 2000; do
  (cd /tmp && rm -rf /tmp/st_tr && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st_tr -- python $R/tools/roi_one.py 99 151 $r 8 > /dev/null 2>&1)
  echo "== R=$r" >> $O/r6_19_st_split.txt
  python tools/prof_summary.py /tmp/st_tr /tmp/st_sum.txt > /dev/null 2>&1; grep -i "roi\|copy\|fill" /tmp/st_sum.txt | cut -c1-200 >> $O/r6_19_st_split.txt
done
cat $O/r6_19_st_split.txt
bash tools/pmc_kernel.sh $O/r6_19_st_pmc.txt '%roi_pool7_st%' python $R/tools/roi_one.py 99 151 2000 4
cat $O/r6_19_st_pmc.txt | cut -c1-700
