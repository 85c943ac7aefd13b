#!/bin/bash
# PMC pass over tools/conv_bench.py for the LDS-resident 3x3 / 64-channel conv kernel (conv3x3_c64_kernel): where do its cycles go?
# usage: tools/pmc_conv_patch.sh <tag>   -> gpurun_out/<tag>_pmc_conv_patch.txt
tag=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
: > $O/${tag}_pmc_conv_patch.txt
for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  d=$O/${tag}_pmcraw_conv
  rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $g -d $d -- python $R/tools/conv_bench.py > /dev/null 2>&1
  python - "$d" >> $O/${tag}_pmc_conv_patch.txt <<'PY'
import glob, sqlite3, sys, collections
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for kn, cn, v, dur, grid in cur.execute("select kernel_name, counter_name, value, duration, grid_size_x from counters_collection where kernel_name like '%conv3x3_c64%'"):
    acc[grid][cn].append(v); acc[grid]["duration_ns"].append(dur)
for grid, c in acc.items():
    print("conv3x3_c64_kernel grid_x=%s: " % grid + "  ".join("%s=%.4g" % (k, sum(v) / len(v)) for k, v in sorted(c.items())))
PY
  rm -rf $d
done
cat $O/${tag}_pmc_conv_patch.txt
