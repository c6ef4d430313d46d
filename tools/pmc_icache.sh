#!/bin/bash
# Instruction-cache / fetch counters for the fused float kernel (run on the GPU box through gpurun).
# usage: pmc_icache.sh MODE TAG
MODE=${1:-0xA1}; TAG=${2:-pmc_ic}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
pass() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/p$n -- python3 tools/run_once.py $MODE 4096 3 > $OUT/p$n.log 2>&1 || echo "pass $n failed"; }
pass 1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES
pass 2 SQC_TC_INST_REQ SQC_TC_STALL SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
python3 tools/pmc_summary.py $OUT
