#!/bin/bash
# PMC attribution passes for the fc6 GEMM family: where do the idle MFMA cycles go?
# One rocprofv3 pass per counter group (SQ has 8 slots, TCC 4, FETCH_SIZE takes 3 of them, WRITE_SIZE 2 -
# MI355X_MICROARCH.md "rocprofv3 PMC slots"); never combined with a trace domain other than --kernel-trace.
# usage: tools/pmc_attrib.sh <tag>        -> gpurun_out/<tag>_pmc_{fwd,dw}.json
tag=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
declare -A PG
PG[sq_time]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
PG[sq_inst]="SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16"
PG[sq_misc]="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT"
PG[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
PG[tcp]="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"
PG[fetch]="FETCH_SIZE"
PG[write]="WRITE_SIZE"
run_shape() {  # name, PMC_SHAPE, bf16-out flag
  name=$1; shape=$2; bf=$3
  for g in sq_time sq_inst sq_misc tcc tcp fetch write; do
    d=$O/${tag}_pmcraw_${name}_${g}
    rm -rf $d
    PMC_SHAPE=$shape PMC_BF16_OUT=$bf timeout 300 rocprofv3 --kernel-trace --pmc ${PG[$g]} -d $d -- python $R/tools/pmc_gemm.py > $O/${tag}_pmc_${name}_${g}.log 2>&1
    echo "$name $g rc=$?"
  done
  PMC_SHAPE=$shape PMC_BF16_OUT=$bf python $R/tools/pmc_attrib_summary.py $O/${tag}_pmcraw_${name} $O/${tag}_pmc_${name}.json
  rm -rf $O/${tag}_pmcraw_${name}_*
}
if [ "${PMC_ONLY:-}" = "dw_tn" ]; then  # the fc6 dW slab reading the pooled matrix K-major (drn_gemm_tn) next to the NT form
  run_shape dw 1024,49152,2048,1 1
  export PMC_TN=1
  run_shape dw_tn 1024,49152,2048,1 1
  exit 0
fi
if [ "${PMC_ONLY:-}" = "sgdp" ]; then  # round 4: the fused fc6 dW + SGD launch (drn_gemm_tn_sgd), all 196 tile columns
  export PMC_SGDP=1
  run_shape dw_sgdp 2048,50176,2048,1 1
  exit 0
fi
run_shape fwd 2000,2048,50176,4 ""
run_shape dw 1024,49152,2048,1 1
