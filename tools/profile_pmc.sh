#!/bin/bash
# Developer helper (GPU box): PMC passes over the headline bench, one counter group per run (rocprofv3 --pmc with
# --kernel-trace only), results under gpurun_out/pmc_<group>/.  Summarise with tools/pmc_summary.py.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() {  # name counters...
  local name=$1; shift
  timeout -k 5 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$name -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/pmc_$name.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$name.log; return 1; }
  echo "pass $name done"
}
run wait SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run valu SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run inst SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU
run mem  SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
