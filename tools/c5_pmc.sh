#!/bin/bash
# Developer helper (GPU box): SQ counter passes over tools/c5_run.py (the fused basis loss kernels at the BASELINE config 5
# shape), one counter group per run; prints per-kernel averages.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/c5pmc_*
run() {
  local name=$1; shift
  timeout -k 5 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/c5pmc_$name -- python3 $R/tools/c5_run.py > $R/gpurun_out/c5pmc_$name.log 2>&1 || { tail -3 $R/gpurun_out/c5pmc_$name.log; return 1; }
}
run wait SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY && \
run valu SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM && \
run inst SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU && \
run mem SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/c5pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "basis_loss" not in k: continue
        acc[k[k.index("basis_loss"):][:28]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]): print("   %-26s %14.4g (x%d)" % (c, sum(acc[k][c]) / len(acc[k][c]), len(acc[k][c])))
PY
