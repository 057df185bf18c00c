#!/bin/bash
# Developer helper: build a VARIANT of libnns_hip.so with extra compile flags for same-box A/B timing.
#   tools/ab_build.sh B "-DNNS_TW_LOOKUP=1"   ->  ab_variants/libnns_hip_B.so     (use with NNS_LIB_PATH=...)
# AB_ONLY="spectral_kernels spectral_bwd_kernels": recompile only these translation units with the flags and link the in-tree objects
# (make -C neural-navier-stokes_amd/csrc first) for the rest -- minutes faster when a macro touches one file.
set -e
TAG=$1; FLAGS=$2
C=neural-navier-stokes_amd/csrc; O=ab_variants; mkdir -p $O/$TAG
HIP="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden $FLAGS"
EXACT="fd_kernels sor_kernels cheb_kernels coarsen_kernels spectral_dense"
EXACT="$EXACT fd_step_kernels"
PLAIN="residual_kernels spectral_kernels spectral_seg_kernels spectral_bwd_kernels neural_kernels pixel_mlp_kernels pixel_mlp_fwd4 spectral_ops slab_kernels pinn_kernels optim_kernels"
want() { [ -z "$AB_ONLY" ] || [[ " $AB_ONLY " == *" $1 "* ]]; }
for f in $EXACT; do if want $f; then $HIP -ffp-contract=off -c $C/$f.hip -o $O/$TAG/$f.o & else cp $C/$f.o $O/$TAG/$f.o; fi; done
for f in $PLAIN; do
  if want $f; then
    noslp=""; case $f in spectral_kernels|spectral_seg_kernels|spectral_bwd_kernels) noslp=-fno-slp-vectorize;; pixel_mlp_fwd4) noslp="-mllvm -amdgpu-mfma-vgpr-form=1";; esac
    $HIP $noslp -c $C/$f.hip -o $O/$TAG/$f.o &
  else cp $C/$f.o $O/$TAG/$f.o; fi
done
if want capi_core; then $HIP -x hip -c $C/capi_core.cpp -o $O/$TAG/capi_core.o & else cp $C/capi_core.o $O/$TAG/capi_core.o; fi
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $O/$TAG/*.o -o $O/libnns_hip_$TAG.so
rm -rf $O/$TAG
echo built $O/libnns_hip_$TAG.so
