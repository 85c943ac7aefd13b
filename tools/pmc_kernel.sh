#!/bin/bash
# Generic PMC attribution of one kernel family: two SQ passes (time split; instruction / LDS mix) and one TCC pass over any command,
# averaged per (kernel, grid) for dispatches whose name matches <like> (SQL LIKE pattern).  Never combined with a trace domain other
# than --kernel-trace.   usage: tools/pmc_kernel.sh <out.txt> <like> <command ...>
out=$1; like=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
: > $out
echo "# command: $*   kernels like: $like" >> $out
for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_SMEM SQ_WAVES" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  d=/tmp/pmc_kernel_raw
  rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $g -d $d -- "$@" > /dev/null 2>&1
  python - "$d" "$like" >> $out <<'PY'
import glob, re, sqlite3, sys, collections
dbs = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)
if not dbs:
    print("pass failed (no database)")
    sys.exit(0)
cur = sqlite3.connect(dbs[0]).cursor()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
q = "select kernel_name, counter_name, value, duration, grid_size_x from counters_collection where kernel_name like ?"
for kn, cn, v, dur, grid in cur.execute(q, (sys.argv[2],)):
    key = (re.sub(r"\(anonymous namespace\)::|void |drn_conv::", "", kn).split("(")[0][:60], grid)
    acc[key][cn].append(v)
    acc[key]["duration_ns"].append(dur)
for (kn, grid), c in sorted(acc.items()):
    n = len(c["duration_ns"])
    avg = {k: sum(v) / len(v) for k, v in c.items()}
    extra = ""
    if "GRBM_GUI_ACTIVE" in avg and avg["GRBM_GUI_ACTIVE"] > 0:
        extra = "  ["
        if avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
            # (GRBM_GUI_ACTIVE spans more than the kernel for short launches: the kernel's own duration at the 2.4-GHz peak clock is
            # the denominator - a lower bound of the busy fraction when the clock is throttled)
            extra += "MFMA pipes busy >= %.3f of 1024 SIMDs x duration x 2.4 GHz, " % (avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * avg["duration_ns"] * 2.4))
        if "SQ_WAVE_CYCLES" in avg and avg["SQ_WAVE_CYCLES"] > 0:
            extra += "waves waiting %.2f / issuing %.2f of their cycles" % (avg.get("SQ_WAIT_INST_ANY", 0) / avg["SQ_WAVE_CYCLES"], avg.get("SQ_ACTIVE_INST_ANY", 0) / avg["SQ_WAVE_CYCLES"])
        extra += "]"
    if "SQ_LDS_IDX_ACTIVE" in avg and avg["SQ_LDS_IDX_ACTIVE"] > 0:
        extra = "  [LDS bank-conflict cycles / LDS active cycles %.2f]" % (avg.get("SQ_LDS_BANK_CONFLICT", 0) / avg["SQ_LDS_IDX_ACTIVE"])
    print("%s grid_x=%s (n=%d): " % (kn, grid, n) + "  ".join("%s=%.5g" % (k, v) for k, v in sorted(avg.items())) + extra)
PY
  rm -rf $d
done
cat $out
