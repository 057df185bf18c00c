#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r2_tests20.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r2_tests20.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r2_smoke.log
bash tools/profile_round.sh r02b
