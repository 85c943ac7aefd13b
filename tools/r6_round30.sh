#!/bin/bash
# evidence for the sparse-table RoIPool: kernel split + PMC attribution at the DC5 map; kernel stats of a DC5 TTA image
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export TMPDIR=/tmp
: > $O/r6_30_st_split.txt
for r in 250 2000; do
  (cd /tmp && rm -rf /tmp/st_tr && ROI_C=2048 ROI_STRIDE=8 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st_tr -- python $R/tools/roi_one.py 99 151 $r 8 > /dev/null 2>&1)
  echo "== 99x151x2048 R=$r (8 launches)" >> $O/r6_30_st_split.txt
  python tools/prof_summary.py /tmp/st_tr /tmp/st_sum.txt > /dev/null 2>&1; grep -i "roi_" /tmp/st_sum.txt | cut -c1-200 >> $O/r6_30_st_split.txt
done
cat $O/r6_30_st_split.txt
ROI_C=2048 ROI_STRIDE=8 bash tools/pmc_kernel.sh $O/r6_30_st_pmc.txt '%roi_pool7_st%' python $R/tools/roi_one.py 99 151 2000 4
cut -c1-900 $O/r6_30_st_pmc.txt
(cd /tmp && rm -rf /tmp/tta_tr && TTA_WORKLOADS=r50dc5 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tta_tr -- python $R/tools/tta_bench.py > $O/r6_30_tta_dc5.txt 2>&1)
python tools/prof_summary.py /tmp/tta_tr $O/r6_30_tta_dc5_kernel_stats.txt > /dev/null 2>&1
head -16 $O/r6_30_tta_dc5_kernel_stats.txt | cut -c1-170
