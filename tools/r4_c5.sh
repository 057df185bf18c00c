#!/bin/bash
# Round 4: the cfg 5 loss + gradient sweep with XOR-swizzled tiles: tests, same-box A/B, counters
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out/r4g
python3 -m pytest tests/test_gpu_neural.py -m gpu -x -q > gpurun_out/r4g/t_neural.log 2>&1 && echo "neural tests ok" || { tail -30 gpurun_out/r4g/t_neural.log; exit 1; }
for round in 1 2 3; do
  for tag in main swz0; do
    if [ "$tag" = main ]; then unset NNS_LIB_PATH; else export NNS_LIB_PATH=$PWD/ab_variants/libnns_hip_$tag.so; fi
    echo "$tag $(python3 tools/c5_time.py 2>/dev/null | tr '\n' ' ')"
  done
done > gpurun_out/r4g/c5_ab.txt 2>&1; cat gpurun_out/r4g/c5_ab.txt
unset NNS_LIB_PATH
PASSES="stats lds wait mfma mem" bash tools/prof_any.sh r4g_c5 tools/mfma_run.py c5
python3 -m pytest tests/test_gpu_residual.py -m gpu -x -q -k "segmented or pack_halo" > gpurun_out/r4g/t_seg.log 2>&1 && echo "seg tests ok" || tail -20 gpurun_out/r4g/t_seg.log
for c in 1 2; do
timeout -k 10 300 python3 bench.py --gpus 1 --mode slab --loopback --chunks $c --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r4g/loopback_c$c.json 2> gpurun_out/r4g/loopback_c$c.err && echo "loopback chunks=$c ok"
done
for round in 1 2 3; do for w in 8 4; do echo "xwaves=$w $(NNS_BWD_XWAVES=$w NNS_PROFILE=1 python3 tools/specbwd_run.py 2>/dev/null)"; done; done > gpurun_out/r4g/specbwd_ab.txt 2>&1; cat gpurun_out/r4g/specbwd_ab.txt
python3 -m pytest tests/test_gpu_residual.py -m gpu -x -q -k "backward or bwd or vjp or adjoint" > gpurun_out/r4g/t_bwd.log 2>&1 && echo "spectral backward tests ok" || tail -20 gpurun_out/r4g/t_bwd.log
echo done
