#!/bin/bash
# Developer helper: build a VARIANT of libnns_hip.so with extra compile flags for same-box A/B timing.
#   tools/ab_build.sh B "-DNNS_TW_LOOKUP=1"   ->  ab_variants/libnns_hip_B.so     (use with NNS_LIB_PATH=...)
set -e
TAG=$1; FLAGS=$2
C=neural-navier-stokes_amd/csrc; O=ab_variants; mkdir -p $O/$TAG
HIP="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden $FLAGS"
for f in fd_kernels sor_kernels cheb_kernels coarsen_kernels; do $HIP -ffp-contract=off -c $C/$f.hip -o $O/$TAG/$f.o & done
for f in residual_kernels spectral_kernels spectral_bwd_kernels neural_kernels pixel_mlp_kernels spectral_ops slab_kernels; do $HIP $( [ ${f#spectral_} != $f -a $f != spectral_ops ] && echo -fno-slp-vectorize ) -c $C/$f.hip -o $O/$TAG/$f.o & done
$HIP -x hip -c $C/capi_core.cpp -o $O/$TAG/capi_core.o &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $O/$TAG/*.o -o $O/libnns_hip_$TAG.so
echo built $O/libnns_hip_$TAG.so
