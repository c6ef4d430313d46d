#!/bin/bash
# PMC passes for the fused float kernel, or the Q15 kernels with MODE = q15 (default cascade) / q15wide (0xA2) (run on the GPU box through gpurun).
# usage: pmc_profile.sh MODE TAG [OUT_KIND]   (SA_PMC_SHORT=1: the SQ and traffic passes only)
# Counters go in separate passes (SQ 8 slots, TCC 4: FETCH_SIZE takes 3, WRITE_SIZE 2).
MODE=${1:-0xA1}; TAG=${2:-pmc}; KIND=${3:-mag_full}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
pass() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/p$n -- python3 tools/run_once.py $MODE 4096 3 $KIND > $OUT/p$n.log 2>&1 || echo "pass $n failed"; }
pass 1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass 2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass 3 FETCH_SIZE GRBM_GUI_ACTIVE
pass 4 WRITE_SIZE TCC_HIT TCC_MISS
[ -n "$SA_PMC_SHORT" ] && { python3 tools/pmc_summary.py $OUT; exit 0; }
pass 5 SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL
pass 6 TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TA_TA_BUSY
python3 tools/pmc_summary.py $OUT
