#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/pm_time.py 2>/dev/null; NNS_LIB_PATH=$PWD/ab_variants/libnns_hip_pmb1.so python tools/pm_time.py 2>/dev/null
python tools/pm_time.py 2>/dev/null; NNS_LIB_PATH=$PWD/ab_variants/libnns_hip_pmb1.so python tools/pm_time.py 2>/dev/null
