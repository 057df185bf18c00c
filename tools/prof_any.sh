#!/bin/bash
# Developer helper (GPU box): the evidence set for ONE python program -- rocprofv3 kernel stats, then the counter passes
# (one counter group per process: HBM traffic, SQ waits / busy, LDS conflicts, instruction mix, matrix-core busy).
#   tools/prof_any.sh TAG tools/specbwd_run.py [args...]     ->  gpurun_out/TAG_stats/, gpurun_out/TAG_pmc_<group>/
# The program comes straight after `--` (python3 itself: no wrapper hop).  PASSES="stats hbm wait lds inst mem mfma" selects.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
PROG=$1; shift
cd /tmp && export TMPDIR=/tmp
export NNS_PROFILE=1              # the run scripts shorten their loops when this is set
mkdir -p $R/gpurun_out
PASSES=${PASSES:-stats hbm wait lds inst mem}
pmc() {  # name counters...
  local name=$1; shift
  timeout -k 5 ${PMC_TIMEOUT:-200} rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_$name -- python3 $R/$PROG $ARGS > $R/gpurun_out/${TAG}_pmc_$name.log 2>&1 || { echo "pass $TAG/$name FAILED"; tail -3 $R/gpurun_out/${TAG}_pmc_$name.log; return 0; }
  echo "pass $TAG/$name done"
}
ARGS="$@"
for p in $PASSES; do
  case $p in
    stats) timeout -k 5 ${PMC_TIMEOUT:-200} rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/$PROG $ARGS > $R/gpurun_out/${TAG}_stats.log 2>&1 || { echo "stats FAILED"; tail -3 $R/gpurun_out/${TAG}_stats.log; } ;;
    hbm)   pmc fetch FETCH_SIZE; pmc write WRITE_SIZE ;;
    wait)  pmc wait SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY ;;
    lds)   pmc lds SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE ;;
    inst)  pmc inst SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU ;;
    mem)   pmc mem SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE ;;
    mfma)  pmc mfma SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES ;;
  esac
done
echo "prof_any $TAG done"
