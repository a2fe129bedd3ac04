#!/bin/bash
# Runs ON the GPU box: SQ / TA / TCP / TCC counter passes over the GEMM variants (tools/gemm_pmc_run.py), one group per pass.
# usage: tools/gpu_gemm_pmc.sh <out dir under gpurun_out>; summary: tools/gemm_pmc_summary.py <dir>
set -u
OUT=gpurun_out/${1:-r04pmc}
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
i=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_VMEM" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" \
  "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_IFETCH" \
  "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TD_TD_BUSY" \
  "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCP_TA_DATA_STALL_CYCLES" \
  "TCC_HIT TCC_MISS TCC_REQ TCC_TAG_STALL" \
  "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$GRAFT_REPO_ROOT/$OUT/pmc_$i" -o run -- \
      python "$GRAFT_REPO_ROOT/tools/gemm_pmc_run.py" ) > "$OUT/pmc_$i.log" 2>&1
  echo "pass $i rc=$? ($grp)"
  find "$OUT/pmc_$i" -name "*.csv" -size +20M -delete
done
du -sh "$OUT"
