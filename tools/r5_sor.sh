#!/bin/bash
# Developer helper (GPU box): the row-per-lane SOR pipeline against the LDS-exchange one -- bitwise on many shapes, the FD tests, then cfg 1 timed with both.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OLD=$R/ab_variants/libnns_hip_sorold.so
timeout -k 10 300 python tools/sor_ab.py gpurun_out/sor_new.npz || exit 1
NNS_LIB_PATH=$OLD timeout -k 10 300 python tools/sor_ab.py gpurun_out/sor_old.npz || exit 1
python tools/sor_ab.py --compare gpurun_out/sor_new.npz gpurun_out/sor_old.npz || exit 1
rm -f gpurun_out/sor_new.npz gpurun_out/sor_old.npz
timeout -k 10 600 python -m pytest tests/test_gpu_fd.py -x -q -m gpu 2>&1 | tail -3 || exit 1
for r in 1 2; do
  echo -n "old  "; NNS_LIB_PATH=$OLD timeout -k 10 100 python tools/c1_run.py
  echo -n "rows "; timeout -k 10 100 python tools/c1_run.py
done
PASSES=stats bash tools/prof_any.sh r04_c1_rows tools/c1_run.py
echo sor done
