#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; timeout -k 5 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pm_$name -- python3 $R/tools/pm_run.py > $R/gpurun_out/pm_$name.log 2>&1 || tail -3 $R/gpurun_out/pm_$name.log; }
run wait SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run valu SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run inst SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU
run mfma SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA
echo done
