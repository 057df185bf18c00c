#!/bin/bash
# Developer helper (GPU box): matrix-core counter passes over the MFMA kernels (pixel MLP forward / backward, the MFMA loss + gradient sweep, the
# ODE MLP kernels): rocprofv3 --pmc with --kernel-trace only, one counter group per process, results under gpurun_out/mfma_<family>_<group>/.
# Summarise with tools/mfma_summary.py -> profiles/rNN_mfma_pmc.csv.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/rocprofv3_counters.txt 2>&1 || true
run() {  # family group counters...
  local fam=$1 name=$2; shift; shift
  timeout -k 5 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/mfma_${fam}_$name -- python3 $R/tools/mfma_run.py $fam > $R/gpurun_out/mfma_${fam}_$name.log 2>&1 || { echo "pass $fam/$name FAILED"; tail -3 $R/gpurun_out/mfma_${fam}_$name.log; return 0; }
  echo "pass $fam/$name done"
}
for fam in ${FAMILIES:-pm c5 ode}; do
  run $fam mops SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
  run $fam busy SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
  run $fam lds  SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  run $fam valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
done
echo done
