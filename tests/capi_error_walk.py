"""Walks the error paths of the C ABI (run by tests/test_capi_errors_asan.py in a child process with the ASan runtime
pre-loaded, against csrc/libnns_hip_asan.so -- the HOST halves of every translation unit built with
-fsanitize=address,undefined).  Every call here fails validation BEFORE any kernel launch, so it is safe with or without a
GPU; a sanitizer report aborts the process (non-zero exit), which fails the test."""
import ctypes as C
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.modules.setdefault('torch', types.ModuleType('torch'))          # nns._lib imports torch only to order the HIP runtime: not needed here
sys.path.insert(0, os.path.join(ROOT, 'neural-navier-stokes_amd'))
os.environ['NNS_LIB_PATH'] = sys.argv[1]
from nns import _lib                                                 # noqa: E402

L = _lib.lib()
OK, INVALID, UNSUPPORTED, LAUNCH, WORKSPACE = 0, -1, -2, -3, -4
seen = set()
buf = (C.c_char * 65536)()                                           # a non-NULL "device" pointer that is never dereferenced on the host
PTR = C.cast(buf, C.c_void_p).value


def expect(name, rc, want):
    msg = L.nns_last_error()
    assert rc in (want if isinstance(want, tuple) else (want,)), '%s returned %d, expected %s (%s)' % (name, rc, want, msg)
    assert msg and len(msg) > 3, '%s: empty nns_last_error after a failure' % name
    seen.add(rc)


def zero_args(argtypes):
    out = []
    for t in argtypes:
        if t is C.c_void_p or t is C.c_char_p or (hasattr(t, '_type_') and not isinstance(t._type_, str)):
            out.append(None)
        elif t in (C.c_double, C.c_float):
            out.append(0.0)
        else:
            out.append(0)
    return out


# ---- 1. every entry point with NULL pointers and zero sizes: a clean INVALID_ARG (or UNSUPPORTED), never a crash
names = dict(_lib._SINGLE)
for base, args in _lib._DUAL.items():
    names[base + '_f32'] = args
    names[base + '_f64'] = args
queries = ('nns_version', 'nns_last_error', 'nns_device_info', 'nns_ode_mlp_bwd_workspace', 'nns_fd_predictor_adi_workspace', 'nns_fd_sor_workspace',
           'nns_fd_sor_redblack_workspace', 'nns_spec_resolve_precise', 'nns_pinn_workspace_bytes', 'nns_fd_step_explicit_fits')
for name, argtypes in sorted(names.items()):
    if name in queries:
        continue
    rc = getattr(L, name)(*zero_args(argtypes))
    expect(name + '(NULL, 0, ...)', rc, (INVALID, UNSUPPORTED))
for q in ('nns_ode_mlp_bwd_workspace', 'nns_fd_predictor_adi_workspace', 'nns_fd_sor_workspace', 'nns_fd_sor_redblack_workspace'):
    assert getattr(L, q)(*zero_args(names[q])) == 0, q              # size queries answer 0 for nonsense
assert L.nns_version() >= 1
assert L.nns_spec_resolve_precise(1, 6.283e-3, 1024, 6.283, 1024, 6.283) == 0 and L.nns_spec_resolve_precise(1, 1.0, 1024, 6.283, 1024, 6.283) == 2     # a policy query
assert L.nns_spec_resolve_precise(0, 1.0, 1024, 6.283, 64, 6.283) == 0 and L.nns_spec_resolve_precise(2, 1e-3, 64, 6.283, 64, 6.283) == 2
rc = L.nns_device_info(None, 0, None, None)
assert rc in (OK, LAUNCH)                                            # LAUNCH without a device: the HIP error string is reported
if rc == LAUNCH:
    seen.add(rc)
    assert L.nns_last_error()

# ---- 2. plausible pointers, arguments that must be refused before any launch
P = PTR
expect('spec_residual nx=5000', L.nns_spec_residual_f32(P, P, P, P, P, P, P, P, 1, 5000, 64, 1e-3, 6.28, 6.28, 1.0, 0.01, 1, None), UNSUPPORTED)
expect('spec_residual_bwd ny=2', L.nns_spec_residual_bwd_f32(*([P] * 10), 1, 64, 2, 1e-3, 6.28, 6.28, 1.0, 0.01, 1, None), UNSUPPORTED)
expect('residual_both ny=4096', L.nns_residual_both_f32(*([P] * 11), 1, 64, 4096, 1e-3, 6.28, 6.28, 1.0, 0.01, 1, None), UNSUPPORTED)
expect('residual_both dt=0', L.nns_residual_both_f32(*([P] * 11), 1, 64, 64, 0.0, 6.28, 6.28, 1.0, 0.01, 1, None), INVALID)
expect('rowpass_halo without halos', L.nns_residual_both_rowpass_halo_f32(P, P, P, P, P, None, None, P, P, P, P, P, P, 1, 8, 64, 0, 1e-3, 0.1, 6.28, 1.0, 0.01, 1, None), INVALID)
expect('rowpass_halo short halo stride', L.nns_residual_both_rowpass_halo_f32(*([P] * 13), 2, 8, 64, 64, 1e-3, 0.1, 6.28, 1.0, 0.01, 1, None), INVALID)
expect('xpass_seg seg_rows=3', L.nns_spec_residual_xpass_seg_f32(P, P, P, P, P, P, 1, 64, 8, 3, 1000, 6.28, 1.0, 0.01, 1, None), INVALID)
expect('xpass_seg short stride', L.nns_spec_residual_xpass_seg_f32(P, P, P, P, P, P, 1, 64, 8, 16, 4, 6.28, 1.0, 0.01, 1, None), INVALID)
expect('ypass_seg seg_cols=24', L.nns_spec_residual_ypass_seg_f32(*([P] * 8), 24, 100000, P, P, P, 1, 8, 64, 1e-3, 6.28, 1.0, 0.01, 1, None), INVALID)
expect('ypass_seg seg_cols < ny / 16', L.nns_spec_residual_ypass_seg_f32(*([P] * 8), 2, 100000, P, P, P, 1, 8, 64, 1e-3, 6.28, 1.0, 0.01, 1, None), INVALID)
expect('ypass_seg short stride', L.nns_spec_residual_ypass_seg_f32(*([P] * 8), 16, 100, P, P, P, 1, 8, 64, 1e-3, 6.28, 1.0, 0.01, 1, None), INVALID)
expect('ypass_seg ny=96', L.nns_spec_residual_ypass_seg_f32(*([P] * 8), 32, 100000, P, P, P, 1, 8, 96, 1e-3, 6.28, 1.0, 0.01, 1, None), UNSUPPORTED)
expect('rowpass_halo_seg no partials', L.nns_residual_both_rowpass_halo_seg_f32(*([P] * 7), None, P, P, 16, 100000, *([P] * 6), 1, 8, 64, 0, 1e-3, 0.1, 6.28, 1.0, 0.01, 1, None), INVALID)
expect('rowpass_halo_seg no halo', L.nns_residual_both_rowpass_halo_seg_f32(*([P] * 5), None, None, P, P, P, 16, 100000, *([P] * 6), 1, 8, 64, 0, 1e-3, 0.1, 6.28, 1.0, 0.01, 1, None), INVALID)
expect('rowpass_halo_seg huge block', L.nns_residual_both_rowpass_halo_seg_f32(*([P] * 10), 1024, 1 << 40, *([P] * 6), 4096, 1024, 1024, 0, 1e-3, 0.1, 6.28, 1.0, 0.01, 1, None), UNSUPPORTED)
expect('fd_residual stencil=7', L.nns_fd_residual_f32(P, P, P, P, P, P, P, P, 1, 8, 8, 1e-3, 0.1, 0.1, 1.0, 0.01, 7, None), INVALID)
expect('fd_residual_halo rows', L.nns_fd_residual_halo_f64(P, P, P, P, P, P, P, P, P, P, 1, 8, 8, 5, 20, 1e-3, 0.1, 0.1, 1.0, 0.01, 5, None), INVALID)
expect('fd_residual_halo no halo', L.nns_fd_residual_halo_f32(P, P, P, P, P, None, None, P, P, P, 1, 8, 8, 0, 8, 1e-3, 0.1, 0.1, 1.0, 0.01, 5, None), INVALID)
expect('basis_loss K=1000', L.nns_basis_loss_fwd_f32(P, P, P, P, 4, 1000, 3, 4096, None), UNSUPPORTED)
expect('basis_loss_fused K=1000', L.nns_basis_loss_fused_f32(P, P, P, P, P, P, 4, 1000, 3, 4096, None), UNSUPPORTED)
expect('ode_mlp hidden=64', L.nns_ode_mlp_fwd_f32(*([P] * 8), 4, 30, 64, 10, 2, None), UNSUPPORTED)
expect('ode_mlp method=9', L.nns_ode_mlp_fwd_f32(*([P] * 8), 4, 30, 128, 10, 9, None), INVALID)
bl = _lib.BcList()
bl.n = 99
expect('bc_apply n=99', L.nns_bc_apply_f32(P, 1, 8, 8, C.byref(bl), None), INVALID)
bl.n = 1
bl.kind[0] = 5
expect('bc_apply kind=5', L.nns_bc_apply_f64(P, 1, 8, 8, C.byref(bl), None), INVALID)
expect('adi nx != ny', L.nns_fd_predictor_adi_f32(*([P] * 7), 1, 8, 9, 1e-3, 0.1, 0.1, 0.1, None), (INVALID, UNSUPPORTED))
expect('sor max_sweeps<0', L.nns_fd_sor_redblack_f64(P, P, P, P, 1, 8, 8, 0.1, 0.1, 1.2, 1e-6, -1, None), INVALID)
expect('halfsweep colour=2', L.nns_fd_sor_redblack_halfsweep_f32(P, P, P, 8, 8, 0, 2, 0.1, 0.1, 1.2, None), INVALID)
expect('halfsweep_gated no prev', L.nns_fd_sor_redblack_halfsweep_gated_f64(P, P, P, None, 1e-6, 8, 8, 0, 0, 0.1, 0.1, 1.2, None), INVALID)
w = (C.c_int * 3)(3, 16, 3)
nb = C.c_size_t(0)
assert L.nns_pixel_mlp_bwd_workspace(w, 2, C.byref(nb)) == 0 and nb.value > 0
expect('pixel_mlp_bwd small workspace', L.nns_pixel_mlp_bwd_f32(*([P] * 7), 1, 4096, w, 2, 1, P, 16, None), WORKSPACE)
wide = (C.c_int * 3)(3, 64, 3)
expect('pixel_mlp_bwd f32 width 64', L.nns_pixel_mlp_bwd_f32(*([P] * 7), 1, 4096, wide, 2, 0, P, 1 << 30, None), UNSUPPORTED)
toow = (C.c_int * 3)(3, 999, 3)
expect('pixel_mlp_fwd width 999', L.nns_pixel_mlp_fwd_f32(P, P, P, P, 1, 4096, toow, 2, 1, None), (INVALID, UNSUPPORTED))
ptrs5 = (C.c_void_p * 5)(*[P] * 5)
expect('slab gather nfields=5', L.nns_slab_gather_lines_f32(ptrs5, 5, P, 1, 0, 0, 8, 1, None), INVALID)
ptrs2 = (C.c_void_p * 2)(P, None)
expect('slab gather NULL field', L.nns_slab_gather_lines_f64(ptrs2, 2, P, 1, 0, 0, 8, 1, None), INVALID)
expect('slab transpose ny % P', L.nns_slab_transpose_pack_f32(ptrs5, 3, P, 1, 4, 10, 4, None), INVALID)
expect('slab pack_halo g0 + Bc > B', L.nns_slab_pack_halo_f32(ptrs5, 3, P, P, P, 4, 3, 2, 8, 64, 4, None), INVALID)
expect('slab pack_halo one halo buffer', L.nns_slab_pack_halo_f64(ptrs5, 3, P, P, None, 4, 0, 2, 8, 64, 4, None), INVALID)
expect('slab pack_halo ny / P not a vector multiple', L.nns_slab_pack_halo_f32(ptrs5, 3, P, None, None, 4, 0, 2, 8, 12, 4, None), UNSUPPORTED)
assert L.nns_pinn_workspace_bytes() >= 4 * 1024 * 8
assert L.nns_fd_step_explicit_fits(64, 64, 8) == 1 and L.nns_fd_step_explicit_fits(200, 200, 8) == 0 and L.nns_fd_step_explicit_fits(64, 64, 2) == 0
expect('pinn assemble one prev field', L.nns_pinn_assemble_f32(P, P, P, P, P, P, P, None, P, 2, 64, None), INVALID)
expect('pinn assemble misaligned workspace', L.nns_pinn_assemble_f32(P, P, P, P, P, P, None, None, P + 4, 2, 64, None), INVALID)
expect('pinn loss n = 0', L.nns_pinn_loss_f32(P, P, P, 0, P, 0.0, 1.0, 1.0, P, None), INVALID)
expect('pinn combine target without fields', L.nns_pinn_combine_f32(P, P, P, None, None, None, P, P, P, 1.0, 1.0, P, 1, 64, None), INVALID)
one = (C.c_void_p * 1)(P)
expect('adam step = 0', L.nns_adam_step_f32(one, one, one, one, (C.c_long * 1)(8), 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, 0, None), INVALID)
expect('adam beta1 = 1', L.nns_adam_step_f32(one, one, one, one, (C.c_long * 1)(8), 1, 1e-3, 1.0, 0.999, 1e-8, 0.0, 1, 0, None), INVALID)
expect('adam NULL state', L.nns_adam_step_f32(one, one, (C.c_void_p * 1)(None), one, (C.c_long * 1)(8), 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 0, None), INVALID)
expect('coarsen agg', L.nns_coarsen_f32(P, P, P, P, P, P, 2, 8, 8, 3, 3, 2, None), (INVALID, UNSUPPORTED))
expect('cheb_gemm M=0', L.nns_cheb_gemm_f64(P, 4, 0, P, 4, 0, P, 4, 0, 4, 4, 1.0, 0.0, 1, None), (INVALID, UNSUPPORTED))
expect('rfft2 nx=48', L.nns_spec_rfft2_f32(P, P, 1, 48, 64, None), (INVALID, UNSUPPORTED))

assert {INVALID, UNSUPPORTED, WORKSPACE} <= seen, seen
print('walked %d entry points; codes seen: %s' % (len(names), sorted(seen)))
