"""ctypes binding of libnns_hip.so (C ABI: include/nns.h).

The HIP library is the product: there is NO CPU fallback.  If the shared object is missing the
import of any op raises with build instructions -- it never degrades to NumPy/PyTorch math.
"""
import ctypes as C
import os

# torch ships its own libamdhip64: it must be loaded BEFORE libnns_hip.so so that both share ONE HIP
# runtime (the SONAME libamdhip64.so.7 then resolves to the copy already mapped); loading ours first
# binds a second runtime that sees no device.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# NNS_LIB_PATH: developer override used for same-box A/B timing of kernel variants (default: the in-tree build)
LIB_PATH = os.environ.get('NNS_LIB_PATH') or os.path.join(os.path.dirname(_HERE), 'csrc', 'libnns_hip.so')
NNS_MAX_BC = 8


class BcList(C.Structure):
    """struct nns_bc_list (include/nns.h)."""
    _fields_ = [('n', C.c_int32), ('kind', C.c_int32 * NNS_MAX_BC), ('side', C.c_int32 * NNS_MAX_BC),
                ('value', C.c_double * NNS_MAX_BC), ('dx', C.c_double * NNS_MAX_BC), ('dy', C.c_double * NNS_MAX_BC)]


_lib = None

_P, _I, _D, _SZ, _L = C.c_void_p, C.c_int, C.c_double, C.c_size_t, C.c_long
_PP = C.POINTER(C.c_void_p)          # host array of device pointers
_BCP = C.POINTER(BcList)

# name -> argtypes (restype is int unless listed in _RESTYPES); the fd/bc ops exist as _f32 and _f64
_DUAL = {
    'nns_bc_apply': [_P, _I, _I, _I, _BCP, _P],
    'nns_fd_predictor_explicit': [_P] * 6 + [_I] * 3 + [_D] * 4 + [_P],
    'nns_fd_predictor_explicit_corrected': [_P] * 6 + [_I] * 3 + [_D] * 4 + [_P],
    'nns_fd_predictor_adi': [_P] * 7 + [_I] * 3 + [_D] * 4 + [_P],
    'nns_fd_predictor_adi_corrected': [_P] * 7 + [_I] * 3 + [_D] * 4 + [_P],
    'nns_fd_predictor_adi_colslab': [_P] * 7 + [_I] * 3 + [_D] * 4 + [_P],
    'nns_fd_pressure_rhs': [_P] * 3 + [_I] * 3 + [_D] * 4 + [_P],
    'nns_fd_sor': [_P] * 4 + [_I] * 3 + [_D] * 4 + [_I, _P],
    'nns_fd_sor_hint': [_P] * 5 + [_I] * 3 + [_D] * 4 + [_I, _P],
    'nns_fd_sor_redblack': [_P] * 4 + [_I] * 3 + [_D] * 4 + [_I, _P],
    'nns_fd_sor_redblack_halfsweep': [_P] * 3 + [_I] * 4 + [_D] * 3 + [_P],
    'nns_fd_correction': [_P] * 5 + [_I] * 3 + [_D] * 3 + [_P],
    'nns_fd_step_explicit': [_P] * 5 + [_BCP] * 3 + [_P] * 6 + [_I] * 3 + [_D] * 7 + [_I, _I, _P],
    'nns_fd_build_b': [_P] * 3 + [_I] * 3 + [_D] * 4 + [_P],
    'nns_fd_jacobi': [_P] * 3 + [_I] * 3 + [_D] * 2 + [_I, _BCP, _P],
    'nns_fd_direct_update': [_P] * 5 + [_I] * 3 + [_D] * 5 + [_P],
    'nns_fd_residual': [_P] * 8 + [_I] * 3 + [_D] * 5 + [_I, _P],
    'nns_fd_residual_bwd': [_P] * 10 + [_I] * 3 + [_D] * 5 + [_I, _P],
    'nns_coarsen': [_P] * 6 + [_I] * 6 + [_P],
    'nns_fd_sor_redblack_halfsweep_gated': [_P] * 4 + [_D] + [_I] * 4 + [_D] * 3 + [_P],
    'nns_fd_residual_halo': [_P] * 10 + [_I] * 5 + [_D] * 5 + [_I, _P],
    'nns_slab_gather_lines': [_PP, _I, _P] + [_L] * 5 + [_P],
    'nns_slab_scatter_lines': [_P, _PP, _I] + [_L] * 5 + [_P],
    'nns_slab_transpose_pack': [_PP, _I, _P] + [_I] * 4 + [_P],
    'nns_slab_transpose_unpack': [_P, _PP, _I] + [_I] * 4 + [_P],
    'nns_slab_pack_halo': [_PP, _I, _P, _P, _P] + [_I] * 6 + [_P],
}
_SINGLE = {
    'nns_spec_residual_f32': [_P] * 8 + [_I] * 3 + [_D] * 5 + [_I, _P],
    'nns_residual_both_f32': [_P] * 11 + [_I] * 3 + [_D] * 5 + [_I, _P],
    'nns_residual_both_rowpass_f32': [_P] * 11 + [_I] * 3 + [_D] * 5 + [_I, _P],
    'nns_spec_residual_bwd_f32': [_P] * 10 + [_I] * 3 + [_D] * 5 + [_I, _P],
    'nns_spec_residual_xpass_f32': [_P] * 6 + [_I] * 3 + [_D] * 3 + [_I, _P],
    'nns_spec_residual_xpass_seg_f32': [_P] * 6 + [_I] * 4 + [_L] + [_D] * 3 + [_I, _P],
    'nns_residual_both_rowpass_halo_f32': [_P] * 13 + [_I] * 3 + [_L] + [_D] * 5 + [_I, _P],
    'nns_spec_residual_ypass_f32': [_P] * 8 + [_I] * 3 + [_D] * 4 + [_I, _P],
    'nns_spec_residual_ypass_seg_f32': [_P] * 8 + [_I, _L] + [_P] * 3 + [_I] * 3 + [_D] * 4 + [_I, _P],
    'nns_residual_both_rowpass_halo_seg_f32': [_P] * 10 + [_I, _L] + [_P] * 6 + [_I] * 3 + [_L] + [_D] * 5 + [_I, _P],
    'nns_spec_resolve_precise': [_I, _D, _I, _D, _I, _D],
    'nns_spec_dense_warmup': [_I, _D],
    'nns_spec_derivs_f32': [_P] * 4 + [_I] * 3 + [_D, _D, _I, _P],
    'nns_spec_rfft2_f32': [_P, _P, _I, _I, _I, _P],
    'nns_spec_irfft2_f32': [_P, _P, _I, _I, _I, _P],
    'nns_pixel_mlp_fwd_f32': [_P] * 4 + [_I, _I, C.POINTER(C.c_int), _I, _I, _P],
    'nns_pixel_mlp_bwd_workspace': [C.POINTER(C.c_int), _I, C.POINTER(C.c_size_t)],
    'nns_pixel_mlp_bwd_f32': [_P] * 7 + [_I, _I, C.POINTER(C.c_int), _I, _I, _P, C.c_size_t, _P],
    'nns_cheb_gemm_f64': [_P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _D, _D, _I, _P],
    'nns_cheb_helmholtz_rhs_f64': [_P] * 12 + [_I, _D, _P],
    'nns_cheb_diag_div_f64': [_P] * 4 + [_I, _I, _D, _D, _D, _P],
    'nns_cheb_embed_f64': [_P] * 6 + [_I, _I, _P],
    'nns_ode_mlp_fwd_f32': [_P] * 8 + [_I] * 5 + [_P],
    'nns_ode_mlp_bwd_f32': [_P] * 17 + [_I] * 5 + [_P],
    'nns_ode_mlp_bwd_workspace': [_I],
    'nns_ode_mlp_bwd_steps_f32': [_P] * 16 + [_I] * 3 + [_D, _I, _P],
    'nns_ode_adjoint_chain_f32': [_P] * 3 + [_I] * 3 + [_P],
    'nns_basis_expand_f32': [_P] * 3 + [_I] * 4 + [_P],
    'nns_basis_expand_bwd_f32': [_P] * 5 + [_I] * 4 + [_P],
    'nns_basis_loss_fwd_f32': [_P] * 4 + [_I] * 4 + [_P],
    'nns_basis_loss_bwd_f32': [_P] * 3 + [C.c_float] + [_P] * 2 + [_I] * 4 + [_P],
    'nns_basis_loss_fused_f32': [_P] * 6 + [_I] * 4 + [_P],
    'nns_fd_step_explicit_fits': [_I, _I, _I],
    'nns_adam_step_f32': [_PP] * 4 + [C.POINTER(C.c_long), _I] + [_D] * 5 + [_L, _I, _P],
    'nns_pinn_workspace_bytes': [],
    'nns_pinn_assemble_f32': [_P] * 9 + [_I, _L, _P],
    'nns_pinn_loss_f32': [_P] * 3 + [_L, _P, _D, _D, _D, _P, _P],
    'nns_pinn_combine_f32': [_P] * 9 + [_D, _D, _P, _I, _L, _P],
    'nns_fd_predictor_adi_workspace': [_I, _I, _I, _I],
    'nns_fd_sor_workspace': [_I, _I, _I, _I],
    'nns_fd_sor_redblack_workspace': [_I, _I, _I, _I, _I],
    'nns_device_info': [C.c_char_p, _I, C.POINTER(_I), C.POINTER(_SZ)],
    'nns_version': [],
    'nns_last_error': [],
}
_RESTYPES = {'nns_pinn_workspace_bytes': _L, 'nns_ode_mlp_bwd_workspace': _SZ, 'nns_fd_predictor_adi_workspace': _SZ, 'nns_fd_sor_workspace': _SZ, 'nns_fd_sor_redblack_workspace': _SZ, 'nns_last_error': C.c_char_p}


def exported_names():
    """Every symbol include/nns.h declares (used by the symbol-export test)."""
    names = list(_SINGLE)
    for base in _DUAL:
        names += [base + '_f32', base + '_f64']
    return sorted(names)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libnns_hip.so not found at %s.\nThe HIP extension is the product and has no CPU fallback: build it with\n"
            "    make -C %s      (hipcc --offload-arch=gfx950)\nor  python -c 'import __graft_entry__ as g; g.build()'"
            % (LIB_PATH, os.path.dirname(LIB_PATH)))
    L = C.CDLL(LIB_PATH)
    for base, args in _DUAL.items():
        for suf in ('_f32', '_f64'):
            f = getattr(L, base + suf)
            f.argtypes, f.restype = args, C.c_int
    for name, args in _SINGLE.items():
        f = getattr(L, name)
        f.argtypes, f.restype = args, _RESTYPES.get(name, C.c_int)
    _lib = L
    return L


class NnsError(RuntimeError):
    pass


class CallRecorder(object):
    """Records the C-ABI calls made while it is active (function object + the exact ctypes arguments, pointer arrays kept alive) so that
    a caller whose arguments do not change from step to step -- the slab pipeline of nns/slab.py on fixed buffers -- can REPLAY them without
    re-validating tensors and rebuilding argument lists in Python (one C call per recorded launch: the host cost of a pipeline stage drops
    from ~40 us to ~5).  Usage: `with rec.stage(key): ...ops calls...`, later `rec.replay(key)`."""

    def __init__(self):
        self.stages = {}
        self._cur = None

    class _Proxy(object):
        def __init__(self, real, rec):
            self._real, self._rec = real, rec

        def __getattr__(self, name):
            f = getattr(self._real, name)
            rec = self._rec

            def call(*args):
                if rec._cur is not None and f.restype is C.c_int and name not in ('nns_version', 'nns_spec_resolve_precise'):
                    rec._cur.append((f, args, name))
                return f(*args)
            return call

    def stage(self, key):
        rec = self

        class _Ctx(object):
            def __enter__(self_):
                global _lib
                lib()
                rec._cur = rec.stages.setdefault(key, [])
                del rec._cur[:]
                self_.saved = _lib
                _lib = CallRecorder._Proxy(self_.saved, rec)

            def __exit__(self_, *exc):
                global _lib
                _lib = self_.saved
                rec._cur = None
                return False
        return _Ctx()

    def replay(self, key):
        for f, args, name in self.stages[key]:
            rc = f(*args)
            if rc != 0:
                check(rc, name)


def check(rc, what):
    if rc != 0:
        msg = lib().nns_last_error()
        raise NnsError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else '?'))


def device_info():
    name = C.create_string_buffer(256)
    cus, mem = _I(0), _SZ(0)
    check(lib().nns_device_info(name, 256, C.byref(cus), C.byref(mem)), 'nns_device_info')
    return dict(name=name.value.decode(), cu_count=cus.value, hbm_bytes=mem.value)
