"""Thin, typed wrappers over the C ABI (include/nns.h) for torch device tensors.

PyTorch is plumbing here (device memory + the current HIP stream); all arithmetic happens in the
hand-written HIP kernels of csrc/.  Every function takes contiguous CUDA(HIP) tensors shaped
[nx, ny] or [batch, nx, ny] in float32 or float64, enqueues ONE stream-ordered call on
torch's current stream and returns its outputs as tensors.  Errors raise nns._lib.NnsError with
the library's message; there is no CPU fallback.
"""
import torch

from . import _lib
from ._lib import BcList, check

KIND = {'dirichlet': 0, 'neumann': 1}
SIDE = {'left': 0, 'right': 1, 'bottom': 2, 'top': 3}



def _prec(precise):
    """The C ABI's `precise` argument: 0 = all-float32 transforms (on differenced lines); 1 / True = the library picks -- all-float32 while
    the viscous amplification nu pi N / (sqrt(3) L) <= 8, else float64 forward transforms; 2 = float64 forward transforms always."""
    return 2 if (precise is not True and precise is not False and int(precise) >= 2) else int(bool(precise))


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _suffix(t):
    if t.dtype == torch.float32:
        return '_f32'
    if t.dtype == torch.float64:
        return '_f64'
    raise TypeError("fields must be float32 or float64, got %s" % t.dtype)


def _dims(t):
    if t.dim() == 2:
        return 1, t.shape[0], t.shape[1]
    if t.dim() == 3:
        return t.shape[0], t.shape[1], t.shape[2]
    raise ValueError("field must be [nx, ny] or [batch, nx, ny], got shape %s" % (tuple(t.shape),))


def _chk(*ts):
    t0 = ts[0]
    for t in ts:
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise TypeError("expected a CUDA/HIP torch tensor (the HIP path has no CPU fallback)")
        if not t.is_contiguous():
            raise ValueError("fields must be C-contiguous")
        if t.dtype != t0.dtype or t.shape != t0.shape or t.device != t0.device:
            raise ValueError("all fields of one call must share dtype, shape and device")
    return _suffix(t0), _dims(t0)


def make_bc_list(bcs):
    """bcs: iterable of objects with .type/.boundary/.value/.dx/.dy (nns.boundary classes or the
    reference's own) or of (kind, side, value, dx, dy) tuples."""
    out = BcList()
    bcs = list(bcs)
    if len(bcs) > _lib.NNS_MAX_BC:
        raise ValueError("at most %d boundary conditions per list" % _lib.NNS_MAX_BC)
    out.n = len(bcs)
    for i, b in enumerate(bcs):
        if isinstance(b, (tuple, list)):
            kind, side, value, dx, dy = b
        else:
            kind, side, value, dx, dy = b.type, b.boundary, b.value, b.dx, b.dy
        out.kind[i], out.side[i] = KIND[kind], SIDE[side]
        out.value[i], out.dx[i], out.dy[i] = float(value), float(dx), float(dy)
    return out


def _call(name, suf, *args):
    check(getattr(_lib.lib(), name + suf)(*args), name + suf)


def _p(t):
    return t.data_ptr()


# ----------------------------------------------------------------------------- boundary
def bc_apply_(A, bcs):
    suf, (B, nx, ny) = _chk(A)
    bl = bcs if isinstance(bcs, BcList) else make_bc_list(bcs)
    _call('nns_bc_apply', suf, _p(A), B, nx, ny, bl, _stream())
    return A


# ----------------------------------------------------------------------------- chorin_fd
def fd_predictor_explicit(un, vn, un1, vn1, dt, dx, dy, nu):
    suf, (B, nx, ny) = _chk(un, vn, un1, vn1)
    ui, vi = torch.empty_like(un), torch.empty_like(vn)
    _call('nns_fd_predictor_explicit', suf, _p(un), _p(vn), _p(un1), _p(vn1), _p(ui), _p(vi), B, nx, ny,
          dt, dx, dy, nu, _stream())
    return ui, vi


def fd_predictor_explicit_corrected(un, vn, un1, vn1, dt, dx, dy, nu):
    """The explicit predictor with the true y-advection (an option of the build, not reference behaviour)."""
    suf, (B, nx, ny) = _chk(un, vn, un1, vn1)
    ui, vi = torch.empty_like(un), torch.empty_like(vn)
    _call('nns_fd_predictor_explicit_corrected', suf, _p(un), _p(vn), _p(un1), _p(vn1), _p(ui), _p(vi), B, nx, ny,
          dt, dx, dy, nu, _stream())
    return ui, vi


def fd_predictor_adi(un, vn, un1, vn1, dt, dx, dy, nu, corrected=False, column_slab=False):
    """corrected=True: the second ADI solve runs along axis 1 (an option of the build; nx != ny allowed).
    column_slab=True: the inputs are a column slab [nx, nyl] of a square grid (nns.slab.SlabChorinFD)."""
    if corrected and column_slab:
        raise ValueError("fd_predictor_adi: column slabs are for the reference ADI (both solves along axis 0)")
    suf, (B, nx, ny) = _chk(un, vn, un1, vn1)
    ui, vi = torch.empty_like(un), torch.empty_like(vn)
    nbytes = _lib.lib().nns_fd_predictor_adi_workspace(B, nx, ny, un.element_size())
    work = torch.empty(nbytes // un.element_size(), dtype=un.dtype, device=un.device)
    _call('nns_fd_predictor_adi_corrected' if corrected else 'nns_fd_predictor_adi_colslab' if column_slab else 'nns_fd_predictor_adi', suf, _p(un), _p(vn), _p(un1), _p(vn1), _p(ui), _p(vi), _p(work), B, nx, ny,
          dt, dx, dy, nu, _stream())
    return ui, vi


def fd_pressure_rhs(ui, vi, dt, dx, dy, rho):
    suf, (B, nx, ny) = _chk(ui, vi)
    C = torch.empty_like(ui)
    _call('nns_fd_pressure_rhs', suf, _p(ui), _p(vi), _p(C), B, nx, ny, dt, dx, dy, rho, _stream())
    return C


def _sor_hint(hint, B, p):
    if hint is None:
        return None
    if not (isinstance(hint, torch.Tensor) and hint.is_cuda and hint.dtype == p.dtype and hint.is_contiguous() and tuple(hint.shape) == (B, 2)):
        raise ValueError("SOR hint: the info tensor [batch, 2] of a previous solve of the same grids (same dtype and device) expected")
    return _p(hint)


def fd_sor_(p, C, dx, dy, beta, tol, max_sweeps, hint=None):
    """In place on p.  Returns the device info tensor [batch, 2] = (sweeps done, last err).  hint: the info of the previous solve of the same
    grids in a time loop (sizes the first speculative batch of sweeps; the result does not depend on it)."""
    suf, (B, nx, ny) = _chk(p, C)
    info = torch.empty(B, 2, dtype=p.dtype, device=p.device)
    nbytes = _lib.lib().nns_fd_sor_workspace(B, nx, ny, p.element_size())
    work = torch.empty(nbytes // p.element_size(), dtype=p.dtype, device=p.device)
    _call('nns_fd_sor_hint', suf, _p(p), _p(C), _p(info), _sor_hint(hint, B, p), _p(work), B, nx, ny, dx, dy, beta, tol, int(max_sweeps), _stream())
    return info


def fd_sor_redblack_(p, C, dx, dy, beta, tol, max_sweeps):
    """Red-black SOR, in place on p (an option of the build).  Returns the device info tensor [batch, 2]."""
    suf, (B, nx, ny) = _chk(p, C)
    info = torch.empty(B, 2, dtype=p.dtype, device=p.device)
    nbytes = _lib.lib().nns_fd_sor_redblack_workspace(B, nx, ny, p.element_size(), int(max_sweeps))     # 0: the grids fit LDS
    work = torch.empty(nbytes // p.element_size(), dtype=p.dtype, device=p.device) if nbytes else None
    _call('nns_fd_sor_redblack', suf, _p(p), _p(C), _p(info), _p(work) if nbytes else None, B, nx, ny, dx, dy, beta, tol,
          int(max_sweeps), _stream())
    return info


def fd_sor_redblack_halfsweep_(p, C, err, gi0, colour, dx, dy, beta):
    """One red-black half-sweep in place on a row slab p [nxl, ny] (rows 0 and nxl-1 = halo / boundary rows).
    err: a zeroed one-element tensor of p's dtype; afterwards err.max-accumulates max|p_new - p_old| (bit pattern)."""
    if p.dim() != 2 or p.shape != C.shape or not p.is_cuda or not p.is_contiguous() or not C.is_contiguous():
        raise ValueError("fd_sor_redblack_halfsweep_: p, C must be contiguous 2-D device tensors of one shape")
    suf = '_f32' if p.dtype == torch.float32 else '_f64'
    _call('nns_fd_sor_redblack_halfsweep', suf, _p(p), _p(C), _p(err), p.shape[0], p.shape[1], int(gi0), int(colour), dx, dy, beta, _stream())
    return err


def fd_sor_redblack_halfsweep_gated_(p, C, err, prev_err, tol, gi0, colour, dx, dy, beta):
    """The half-sweep as one link of a pre-enqueued chain (nns_fd_sor_redblack_halfsweep_gated_*): it runs only if the
    one-element device tensor prev_err (the previous sweep's error, already max-reduced over the ranks) is > tol."""
    if p.dim() != 2 or p.shape != C.shape or not p.is_cuda or not p.is_contiguous() or not C.is_contiguous():
        raise ValueError("fd_sor_redblack_halfsweep_gated_: p, C must be contiguous 2-D device tensors of one shape")
    suf = '_f32' if p.dtype == torch.float32 else '_f64'
    _call('nns_fd_sor_redblack_halfsweep_gated', suf, _p(p), _p(C), _p(err), _p(prev_err), float(tol), p.shape[0], p.shape[1], int(gi0), int(colour),
          dx, dy, beta, _stream())
    return err


def fd_correction(ui, vi, p, dt, dx, dy):
    suf, (B, nx, ny) = _chk(ui, vi, p)
    u, v = torch.empty_like(ui), torch.empty_like(vi)
    _call('nns_fd_correction', suf, _p(ui), _p(vi), _p(p), _p(u), _p(v), B, nx, ny, dt, dx, dy, _stream())
    return u, v


def fd_step_explicit_fits(nx, ny, dtype):
    """Does the one-launch explicit step apply to this grid (p and its right-hand side in one workgroup's LDS)?"""
    return bool(_lib.lib().nns_fd_step_explicit_fits(int(nx), int(ny), 8 if dtype == torch.float64 else 4))


def fd_step_explicit(un, vn, un1, vn1, p, u_bcl, v_bcl, p_bcl, dt, dx, dy, rho, nu, beta, tol, max_sweeps, corrected=False, out=None, p_copy=None, hint=None):
    """chorin_fd's explicit step in ONE launch (nns_fd_step_explicit_*): predictor, boundary lists, pressure solve, correction.  p is updated in
    place (and copied to p_copy when given); returns (u, v, info) with u, v = `out` (two fields that are not inputs) or new tensors."""
    suf, (B, nx, ny) = _chk(un, vn, un1, vn1, p)
    u, v = out if out is not None else (torch.empty_like(un), torch.empty_like(un))
    _chk(un, u, v)
    if p_copy is not None:
        _chk(un, p_copy)
    info = torch.empty(B, 2, dtype=p.dtype, device=p.device)
    nbytes = _lib.lib().nns_fd_sor_workspace(B, nx, ny, p.element_size())
    work = torch.empty(nbytes // p.element_size(), dtype=p.dtype, device=p.device)
    _call('nns_fd_step_explicit', suf, _p(un), _p(vn), _p(un1), _p(vn1), _p(p), u_bcl, v_bcl, p_bcl, _p(u), _p(v), _p(p_copy) if p_copy is not None else None,
          _p(info), _sor_hint(hint, B, p), _p(work), B, nx, ny, dt, dx, dy, rho, nu, beta, tol, int(max_sweeps), int(bool(corrected)), _stream())
    return u, v, info


def coarsen(u, v, p, agg_x, agg_y, jfill=None):
    """Block means of the [T, nx, ny] device sequences u, v, p over agg_x x agg_y cells (nns_coarsen_*).
    jfill: coarse columns filled per row (the rest are 0); default ny / agg_y."""
    if not (u.dim() == 3 and u.shape == v.shape == p.shape and u.dtype == v.dtype == p.dtype and u.is_cuda and v.is_cuda and p.is_cuda):
        raise ValueError("coarsen: u, v, p must be device tensors of one [T, nx, ny] shape and dtype")
    if u.dtype not in (torch.float32, torch.float64):
        raise TypeError("coarsen: float32 or float64 fields")
    if not (u.is_contiguous() and v.is_contiguous() and p.is_contiguous()):
        raise ValueError("coarsen: contiguous tensors required")
    T, nx, ny = u.shape
    if agg_x < 1 or agg_y < 1 or nx % agg_x or ny % agg_y:
        raise ValueError("coarsen: nx=%d, ny=%d must be multiples of agg_x=%d, agg_y=%d" % (nx, ny, agg_x, agg_y))
    out = [torch.empty(T, nx // agg_x, ny // agg_y, dtype=u.dtype, device=u.device) for _ in range(3)]
    suf = '_f32' if u.dtype == torch.float32 else '_f64'
    _call('nns_coarsen', suf, _p(u), _p(v), _p(p), _p(out[0]), _p(out[1]), _p(out[2]), T, nx, ny, int(agg_x), int(agg_y),
          ny // agg_y if jfill is None else int(jfill), _stream())
    return tuple(out)


# ----------------------------------------------------------------------------- direct_fd
def fd_build_b(u, v, dt, dx, dy, rho):
    suf, (B, nx, ny) = _chk(u, v)
    b = torch.empty_like(u)
    _call('nns_fd_build_b', suf, _p(u), _p(v), _p(b), B, nx, ny, dt, dx, dy, rho, _stream())
    return b


def fd_jacobi_(p, b, dx, dy, nit, p_bc):
    suf, (B, nx, ny) = _chk(p, b)
    bl = p_bc if isinstance(p_bc, BcList) else make_bc_list(p_bc)
    tmp = torch.empty_like(p)
    _call('nns_fd_jacobi', suf, _p(p), _p(tmp), _p(b), B, nx, ny, dx, dy, int(nit), bl, _stream())
    return p


def fd_direct_update(un, vn, p, dt, dx, dy, rho, nu):
    suf, (B, nx, ny) = _chk(un, vn, p)
    u, v = torch.empty_like(un), torch.empty_like(vn)
    _call('nns_fd_direct_update', suf, _p(un), _p(vn), _p(p), _p(u), _p(v), B, nx, ny, dt, dx, dy, rho, nu, _stream())
    return u, v


# ----------------------------------------------------------------------------- periodic residual
def fd_residual(u, v, p, u_prev, v_prev, dt, dx, dy, rho, nu, stencil=5, out=None):
    suf, (B, nx, ny) = _chk(u, v, p, u_prev, v_prev)
    ru, rv, rd = out if out is not None else (torch.empty_like(u), torch.empty_like(u), torch.empty_like(u))
    _call('nns_fd_residual', suf, _p(u), _p(v), _p(p), _p(u_prev), _p(v_prev), _p(ru), _p(rv), _p(rd), B, nx, ny,
          dt, dx, dy, rho, nu, int(stencil), _stream())
    return ru, rv, rd


def fd_residual_halo(u, v, p, u_prev, v_prev, halo_top, halo_bot, dt, dx, dy, rho, nu, stencil=5, rows=None, out=None):
    """fd_residual on local rows `rows` = (begin, end) of a row slab [B, nloc, ny] whose rows -1 / nloc are the
    [3, B, ny] messages halo_top / halo_bot (nns_fd_residual_halo_*)."""
    suf, (B, nx, ny) = _chk(u, v, p, u_prev, v_prev)
    for h in (halo_top, halo_bot):
        if not (h.is_cuda and h.is_contiguous() and h.dtype == u.dtype and tuple(h.shape) == (3, B, ny)):
            raise ValueError("fd_residual_halo: halo messages must be contiguous [3, %d, %d] device tensors of the fields' dtype" % (B, ny))
    ru, rv, rd = out if out is not None else (torch.empty_like(u), torch.empty_like(u), torch.empty_like(u))
    r0, r1 = rows if rows is not None else (0, nx)
    _call('nns_fd_residual_halo', suf, _p(u), _p(v), _p(p), _p(u_prev), _p(v_prev), _p(halo_top), _p(halo_bot), _p(ru), _p(rv), _p(rd),
          B, nx, ny, int(r0), int(r1), dt, dx, dy, rho, nu, int(stencil), _stream())
    return ru, rv, rd


def fd_residual_bwd(u, v, g_u, g_v, g_div, dt, dx, dy, rho, nu, stencil=5, want_prev=True):
    """Vector-Jacobian product of fd_residual: returns (grad_u, grad_v, grad_p, grad_u_prev, grad_v_prev)
    (the last two None unless want_prev)."""
    suf, (B, nx, ny) = _chk(u, v, g_u, g_v, g_div)
    gu, gv, gp = torch.empty_like(u), torch.empty_like(u), torch.empty_like(u)
    gup, gvp = (torch.empty_like(u), torch.empty_like(u)) if want_prev else (None, None)
    _call('nns_fd_residual_bwd', suf, _p(u), _p(v), _p(g_u), _p(g_v), _p(g_div), _p(gu), _p(gv), _p(gp),
          _p(gup) if want_prev else None, _p(gvp) if want_prev else None, B, nx, ny, dt, dx, dy, rho, nu, int(stencil), _stream())
    return gu, gv, gp, gup, gvp


class FdResidualFn(torch.autograd.Function):
    """fd_residual as an autograd node (forward: nns_fd_residual, backward: nns_fd_residual_bwd, the adjoint stencils)."""

    @staticmethod
    def forward(ctx, u, v, p, u_prev, v_prev, dt, dx, dy, rho, nu, stencil):
        u, v, p, u_prev, v_prev = (t.contiguous() for t in (u, v, p, u_prev, v_prev))
        ctx.save_for_backward(u, v)
        ctx.consts = (dt, dx, dy, rho, nu, stencil)
        return fd_residual(u, v, p, u_prev, v_prev, dt, dx, dy, rho, nu, stencil)

    @staticmethod
    def backward(ctx, g_u, g_v, g_d):
        u, v = ctx.saved_tensors
        dt, dx, dy, rho, nu, stencil = ctx.consts
        zero = lambda g: torch.zeros_like(u) if g is None else g.contiguous()
        want_prev = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        gu, gv, gp, gup, gvp = fd_residual_bwd(u, v, zero(g_u), zero(g_v), zero(g_d), dt, dx, dy, rho, nu, stencil, want_prev)
        return (gu, gv, gp, gup, gvp) + (None,) * 6


def spec_residual(u, v, p, u_prev, v_prev, dt, Lx, Ly, rho, nu, precise=True, out=None):
    suf, (B, nx, ny) = _chk(u, v, p, u_prev, v_prev)
    if suf != '_f32':
        raise TypeError("spec_residual: float32 fields (the forward transforms run in float64 internally)")
    ru, rv, rd = out if out is not None else (torch.empty_like(u), torch.empty_like(u), torch.empty_like(u))
    check(_lib.lib().nns_spec_residual_f32(_p(u), _p(v), _p(p), _p(u_prev), _p(v_prev), _p(ru), _p(rv), _p(rd), B, nx, ny,
                                           dt, Lx, Ly, rho, nu, _prec(precise), _stream()), 'nns_spec_residual_f32')
    return ru, rv, rd


def residual_both(u, v, p, u_prev, v_prev, dt, Lx, Ly, rho, nu, precise=True, out_fd=None, out_spec=None, rowpass_only=False):
    """FD 5-point AND spectral residual of the same inputs (the 'stencil + spectral residual' of BASELINE.json) by
    nns_residual_both_f32: spectral column pass, then ONE row pass that also evaluates the stencil.
    Returns ((fd r_u, r_v, r_div), (spectral r_u, r_v, r_div)).  rowpass_only: out_spec already holds the column pass's
    partials (spec_residual_xpass) and only the second launch runs."""
    suf, (B, nx, ny) = _chk(u, v, p, u_prev, v_prev)
    if suf != '_f32':
        raise TypeError("residual_both: float32 fields")
    fo = out_fd if out_fd is not None else tuple(torch.empty_like(u) for _ in range(3))
    so = out_spec if out_spec is not None else tuple(torch.empty_like(u) for _ in range(3))
    fn = _lib.lib().nns_residual_both_rowpass_f32 if rowpass_only else _lib.lib().nns_residual_both_f32
    check(fn(_p(u), _p(v), _p(p), _p(u_prev), _p(v_prev), _p(fo[0]), _p(fo[1]), _p(fo[2]), _p(so[0]), _p(so[1]), _p(so[2]), B, nx, ny,
             dt, Lx, Ly, rho, nu, _prec(precise), _stream()), 'nns_residual_both_f32')
    return fo, so


def spec_residual_xpass(u, v, p, Lx, rho, nu, precise=True, out=None):
    suf, (B, nx, ny) = _chk(u, v, p)
    if suf != '_f32':
        raise TypeError("spec_residual_xpass: float32 fields")
    ru, rv, rd = out if out is not None else (torch.empty_like(u), torch.empty_like(u), torch.empty_like(u))
    check(_lib.lib().nns_spec_residual_xpass_f32(_p(u), _p(v), _p(p), _p(ru), _p(rv), _p(rd), B, nx, ny, Lx, rho, nu,
                                                 _prec(precise), _stream()), 'nns_spec_residual_xpass_f32')
    return ru, rv, rd


def spec_residual_xpass_seg(recv, send, B, nx, nyl, seg_rows, Lx, rho, nu, precise=True):
    """The column pass on the RECEIVE buffer of the slab all-to-all, recv [P, 3, B, seg_rows, nyl] (u, v, p from every source
    rank), writing its three partials into `send` of the same layout (the send buffer of the return all-to-all):
    nns_spec_residual_xpass_seg_f32."""
    _f32(recv, send)
    if recv.shape != send.shape or recv.dim() != 5 or recv.shape[1] != 3 or recv.shape[0] * seg_rows != nx or tuple(recv.shape[2:]) != (B, seg_rows, nyl):
        raise ValueError("spec_residual_xpass_seg: buffers must be [P, 3, %d, %d, %d] with P * seg_rows = nx = %d, got %s" % (B, seg_rows, nyl, nx, tuple(recv.shape)))
    fs = B * seg_rows * nyl                  # one field of one source rank
    e = recv.element_size()
    q = lambda t, f: t.data_ptr() + f * fs * e
    check(_lib.lib().nns_spec_residual_xpass_seg_f32(q(recv, 0), q(recv, 1), q(recv, 2), q(send, 0), q(send, 1), q(send, 2), B, nx, nyl,
                                                     int(seg_rows), 3 * fs, Lx, rho, nu, _prec(precise), _stream()), 'nns_spec_residual_xpass_seg_f32')
    return send


def residual_both_rowpass_halo(u, v, p, u_prev, v_prev, halo_top, halo_bot, sp_partials, dt, dx, Ly, rho, nu, precise=True, out_fd=None, halo_grid0=0):
    """The fused row pass on a row slab (nns_residual_both_rowpass_halo_f32): sp_partials holds the column pass's partials on
    entry and the spectral residual on return; returns ((fd r_u, r_v, r_div), sp_partials).  halo_top / halo_bot: [3, Bh, ny] messages;
    Bh may exceed this call's batch B -- the call then covers grids halo_grid0 .. halo_grid0 + B - 1 of the batch the messages were
    exchanged for (nns/slab.py: one halo exchange per step, batch chunks pipelined)."""
    suf, (B, nx, ny) = _chk(u, v, p, u_prev, v_prev, *sp_partials)
    if suf != '_f32':
        raise TypeError("residual_both_rowpass_halo: float32 fields")
    Bh = halo_top.shape[1] if halo_top.dim() == 3 else -1
    for h in (halo_top, halo_bot):
        if not (h.is_cuda and h.is_contiguous() and h.dtype == u.dtype and tuple(h.shape) == (3, Bh, ny) and 0 <= halo_grid0 and halo_grid0 + B <= Bh):
            raise ValueError("residual_both_rowpass_halo: halo messages must be contiguous [3, >= %d, %d] float32 device tensors" % (halo_grid0 + B, ny))
    fo = out_fd if out_fd is not None else tuple(torch.empty_like(u) for _ in range(3))
    so = sp_partials
    off = halo_grid0 * ny * 4
    check(_lib.lib().nns_residual_both_rowpass_halo_f32(_p(u), _p(v), _p(p), _p(u_prev), _p(v_prev), _p(halo_top) + off, _p(halo_bot) + off,
                                                        _p(fo[0]), _p(fo[1]), _p(fo[2]), _p(so[0]), _p(so[1]), _p(so[2]), B, nx, ny, Bh * ny,
                                                        dt, dx, Ly, rho, nu, _prec(precise), _stream()), 'nns_residual_both_rowpass_halo_f32')
    return fo, so


def _seg_parts(got, B, nx, ny, what):
    """got [P, 3, B, nx, ny / P] (the receive buffer of the return all-to-all) -> (the three fields' pointers of source rank 0, seg_cols, seg_stride)."""
    _f32(got)
    P = got.shape[0] if got.dim() == 5 else 0
    if P < 1 or ny % P or tuple(got.shape) != (P, 3, B, nx, ny // P):
        raise ValueError("%s: partials must be [P, 3, %d, %d, ny / P] float32, got %s" % (what, B, nx, tuple(got.shape)))
    fs = B * nx * (ny // P)
    return [got.data_ptr() + f * fs * 4 for f in range(3)], ny // P, 3 * fs


def residual_both_rowpass_halo_seg(u, v, p, u_prev, v_prev, halo_top, halo_bot, got, dt, dx, Ly, rho, nu, precise=True, out_fd=None, out_spec=None, halo_grid0=0):
    """The fused row pass on a row slab with its column-pass partials read IN PLACE from `got` [P, 3, B, nloc, ny / P], the receive buffer
    of the slab step's return all-to-all (nns_residual_both_rowpass_halo_seg_f32): no scatter copy in between.  Returns
    ((fd r_u, r_v, r_div), (spectral r_u, r_v, r_div)), both as row slabs.  halo_top / halo_bot / halo_grid0 as residual_both_rowpass_halo."""
    suf, (B, nx, ny) = _chk(u, v, p, u_prev, v_prev)
    if suf != '_f32':
        raise TypeError("residual_both_rowpass_halo_seg: float32 fields")
    Bh = halo_top.shape[1] if halo_top.dim() == 3 else -1
    for h in (halo_top, halo_bot):
        if not (h.is_cuda and h.is_contiguous() and h.dtype == u.dtype and tuple(h.shape) == (3, Bh, ny) and 0 <= halo_grid0 and halo_grid0 + B <= Bh):
            raise ValueError("residual_both_rowpass_halo_seg: halo messages must be contiguous [3, >= %d, %d] float32 device tensors" % (halo_grid0 + B, ny))
    parts, seg_cols, seg_stride = _seg_parts(got, B, nx, ny, 'residual_both_rowpass_halo_seg')
    fo = out_fd if out_fd is not None else tuple(torch.empty_like(u) for _ in range(3))
    so = out_spec if out_spec is not None else tuple(torch.empty_like(u) for _ in range(3))
    _chk(u, *fo, *so)
    off = halo_grid0 * ny * 4
    check(_lib.lib().nns_residual_both_rowpass_halo_seg_f32(_p(u), _p(v), _p(p), _p(u_prev), _p(v_prev), _p(halo_top) + off, _p(halo_bot) + off,
                                                            parts[0], parts[1], parts[2], seg_cols, seg_stride,
                                                            _p(fo[0]), _p(fo[1]), _p(fo[2]), _p(so[0]), _p(so[1]), _p(so[2]), B, nx, ny, Bh * ny,
                                                            dt, dx, Ly, rho, nu, _prec(precise), _stream()), 'nns_residual_both_rowpass_halo_seg_f32')
    return fo, so


def spec_residual_ypass_seg(u, v, p, u_prev, v_prev, got, dt, Ly, rho, nu, precise=True, out=None):
    """The spectral row pass with its partials read in place from `got` [P, 3, B, nloc, ny / P] (nns_spec_residual_ypass_seg_f32)."""
    suf, (B, nx, ny) = _chk(u, v, p, u_prev, v_prev)
    if suf != '_f32':
        raise TypeError("spec_residual_ypass_seg: float32 fields")
    parts, seg_cols, seg_stride = _seg_parts(got, B, nx, ny, 'spec_residual_ypass_seg')
    ro = out if out is not None else tuple(torch.empty_like(u) for _ in range(3))
    _chk(u, *ro)
    check(_lib.lib().nns_spec_residual_ypass_seg_f32(_p(u), _p(v), _p(p), _p(u_prev), _p(v_prev), parts[0], parts[1], parts[2], seg_cols, seg_stride,
                                                     _p(ro[0]), _p(ro[1]), _p(ro[2]), B, nx, ny, dt, Ly, rho, nu, _prec(precise), _stream()),
          'nns_spec_residual_ypass_seg_f32')
    return ro


def spec_resolve_precise(precise, nu, nx, Lx, ny, Ly):
    """The arithmetic a `precise` request resolves to for a whole evaluation: 0 or 2 (nns_spec_resolve_precise: the library's one policy,
    NNS_SPEC_F64 included)."""
    return int(_lib.lib().nns_spec_resolve_precise(_prec(precise), float(nu), int(nx), float(Lx), int(ny), float(Ly)))


# ----------------------------------------------------------------------------- slab message packing (nns/slab.py)
def _ptr_array(ts):
    import ctypes
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def _slab_fields(fields, what):
    t0 = fields[0]
    if not 1 <= len(fields) <= 4:
        raise ValueError("%s: 1 to 4 fields per message" % what)
    for t in fields:
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous() and t.dtype == t0.dtype and t.shape == t0.shape):
            raise ValueError("%s: fields must be contiguous device tensors of one shape and dtype" % what)
    return _suffix(t0)


def slab_gather_lines(fields, msg, nouter, outer_stride, line_off, length, elem_stride=1):
    """msg[f, o, e] = fields[f].flat[o * outer_stride + line_off + e * elem_stride] (one launch, nns_slab_gather_lines_*)."""
    suf = _slab_fields(fields, 'slab_gather_lines')
    if not (msg.is_cuda and msg.is_contiguous() and msg.dtype == fields[0].dtype and msg.numel() == len(fields) * nouter * length):
        raise ValueError("slab_gather_lines: msg must hold nfields * nouter * len elements")
    if (nouter - 1) * outer_stride + line_off + (length - 1) * elem_stride >= fields[0].numel():
        raise ValueError("slab_gather_lines: line runs past the end of the field")
    _call('nns_slab_gather_lines', suf, _ptr_array(fields), len(fields), _p(msg), nouter, outer_stride, line_off, length, elem_stride, _stream())
    return msg


def slab_scatter_lines(msg, fields, nouter, outer_stride, line_off, length, elem_stride=1):
    """The reverse of slab_gather_lines: writes the message's lines into the fields (nns_slab_scatter_lines_*)."""
    suf = _slab_fields(fields, 'slab_scatter_lines')
    if not (msg.is_cuda and msg.is_contiguous() and msg.dtype == fields[0].dtype and msg.numel() == len(fields) * nouter * length):
        raise ValueError("slab_scatter_lines: msg must hold nfields * nouter * len elements")
    if (nouter - 1) * outer_stride + line_off + (length - 1) * elem_stride >= fields[0].numel():
        raise ValueError("slab_scatter_lines: line runs past the end of the field")
    _call('nns_slab_scatter_lines', suf, _p(msg), _ptr_array(fields), len(fields), nouter, outer_stride, line_off, length, elem_stride, _stream())
    return fields


def slab_transpose_pack(fields, send, P):
    """Row slabs fields[f] [B, nloc, ny] -> send [P, F, B, nloc, ny / P] (nns_slab_transpose_pack_*)."""
    suf = _slab_fields(fields, 'slab_transpose_pack')
    B, nloc, ny = fields[0].shape
    if not (send.is_cuda and send.is_contiguous() and send.dtype == fields[0].dtype and tuple(send.shape) == (P, len(fields), B, nloc, ny // P) and ny % P == 0):
        raise ValueError("slab_transpose_pack: send must be [%d, %d, %d, %d, %d]" % (P, len(fields), B, nloc, ny // P))
    _call('nns_slab_transpose_pack', suf, _ptr_array(fields), len(fields), _p(send), B, nloc, ny, P, _stream())
    return send


def slab_pack_halo(fields, send, first, last, g0, P):
    """transpose_pack of grids [g0, g0 + Bc) of the row slabs fields[f] [B, nloc, ny] into send [P, F, Bc, nloc, ny / P] and -- first / last not
    None -- the edge rows of ALL B grids into first / last [F, B, ny], in ONE launch (nns_slab_pack_halo_*)."""
    suf = _slab_fields(fields, 'slab_pack_halo')
    B, nloc, ny = fields[0].shape
    F = len(fields)
    Bc = send.shape[2] if send.dim() == 5 else -1
    if not (send.is_cuda and send.is_contiguous() and send.dtype == fields[0].dtype and ny % P == 0 and tuple(send.shape) == (P, F, Bc, nloc, ny // P) and 0 <= g0 and g0 + Bc <= B):
        raise ValueError("slab_pack_halo: send must be [%d, %d, Bc, %d, %d] with g0 + Bc <= %d" % (P, F, nloc, ny // P, B))
    for h in (first, last):
        if h is not None and not (h.is_cuda and h.is_contiguous() and h.dtype == fields[0].dtype and tuple(h.shape) == (F, B, ny)):
            raise ValueError("slab_pack_halo: first / last must be contiguous [%d, %d, %d]" % (F, B, ny))
    _call('nns_slab_pack_halo', suf, _ptr_array(fields), F, _p(send), _p(first) if first is not None else None, _p(last) if last is not None else None,
          B, int(g0), Bc, nloc, ny, P, _stream())
    return send


def slab_transpose_unpack(recv, fields, P):
    """recv [P, F, B, nloc, ny / P] -> row slabs fields[f] [B, nloc, ny] (nns_slab_transpose_unpack_*)."""
    suf = _slab_fields(fields, 'slab_transpose_unpack')
    B, nloc, ny = fields[0].shape
    if not (recv.is_cuda and recv.is_contiguous() and recv.dtype == fields[0].dtype and tuple(recv.shape) == (P, len(fields), B, nloc, ny // P) and ny % P == 0):
        raise ValueError("slab_transpose_unpack: recv must be [%d, %d, %d, %d, %d]" % (P, len(fields), B, nloc, ny // P))
    _call('nns_slab_transpose_unpack', suf, _p(recv), _ptr_array(fields), len(fields), B, nloc, ny, P, _stream())
    return fields


def spec_residual_ypass_(u, v, p, u_prev, v_prev, ru, rv, rd, dt, Ly, rho, nu, precise=True):
    suf, (B, nx, ny) = _chk(u, v, p, u_prev, v_prev, ru, rv, rd)
    if suf != '_f32':
        raise TypeError("spec_residual_ypass: float32 fields")
    check(_lib.lib().nns_spec_residual_ypass_f32(_p(u), _p(v), _p(p), _p(u_prev), _p(v_prev), _p(ru), _p(rv), _p(rd), B, nx, ny,
                                                 dt, Ly, rho, nu, _prec(precise), _stream()), 'nns_spec_residual_ypass_f32')
    return ru, rv, rd


# ----------------------------------------------------------------------------- neural_spectral
ODE_METHODS = {'Euler': 0, 'RK2': 1, 'RK4': 2}


def _f32(*ts):
    for t in ts:
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise TypeError("expected contiguous float32 CUDA/HIP tensors")


def ode_mlp_fwd(z0, W0, b0, W1, b1, W2, b2, Nt, method):
    _f32(z0, W0, b0, W1, b1, W2, b2)
    mb, K = z0.shape
    out = torch.empty(Nt, mb, K, dtype=torch.float32, device=z0.device)
    check(_lib.lib().nns_ode_mlp_fwd_f32(_p(z0), _p(W0), _p(b0), _p(W1), _p(b1), _p(W2), _p(b2), _p(out), mb, K, W1.shape[0],
                                         int(Nt), ODE_METHODS[method], _stream()), 'nns_ode_mlp_fwd_f32')
    return out


def ode_mlp_bwd(z0, W0, b0, W1, b1, W2, b2, states, grad_out, Nt, method):
    _f32(z0, W0, b0, W1, b1, W2, b2, states, grad_out)
    mb, K = z0.shape
    gz0 = torch.empty_like(z0)
    gs = [torch.empty_like(t) for t in (W0, b0, W1, b1, W2, b2)]
    work = torch.empty(_lib.lib().nns_ode_mlp_bwd_workspace(mb) // 4, dtype=torch.float32, device=z0.device)
    check(_lib.lib().nns_ode_mlp_bwd_f32(_p(z0), _p(W0), _p(b0), _p(W1), _p(b1), _p(W2), _p(b2), _p(states), _p(grad_out), _p(gz0),
                                         *[_p(g) for g in gs], _p(work), mb, K, W1.shape[0], int(Nt), ODE_METHODS[method], _stream()),
          'nns_ode_mlp_bwd_f32')
    return gz0, gs


def ode_mlp_bwd_steps(y, W0, b0, W1, b1, W2, b2, grad_out, dt, method, want_param_grads=True):
    """Backward of rows INDEPENDENT single steps of size dt (nns_ode_mlp_bwd_steps_f32): returns (grad_y [rows, K], parameter
    gradients summed over the rows -- None with want_param_grads=False: the kernel then skips their products, memsets and atomics)."""
    _f32(y, W0, b0, W1, b1, W2, b2, grad_out)
    rows, K = y.shape
    gy = torch.empty_like(y)
    gs = [torch.empty_like(t) for t in (W0, b0, W1, b1, W2, b2)] if want_param_grads else None
    work = torch.empty(_lib.lib().nns_ode_mlp_bwd_workspace(rows) // 4, dtype=torch.float32, device=y.device)
    check(_lib.lib().nns_ode_mlp_bwd_steps_f32(_p(y), _p(W0), _p(b0), _p(W1), _p(b1), _p(W2), _p(b2), _p(grad_out), _p(gy),
                                               *([_p(g) for g in gs] if want_param_grads else [None] * 6), _p(work), rows, K, W1.shape[0], float(dt),
                                               ODE_METHODS[method], _stream()),
          'nns_ode_mlp_bwd_steps_f32')
    return gy, gs


def ode_adjoint_chain(J, g):
    """lam[Nt-1] = g[Nt-1], lam[s-1] = g[s-1] + lam[s] @ J[s] for J [Nt, mb, K, K], g [Nt, mb, K] (nns_ode_adjoint_chain_f32)."""
    _f32(J, g)
    Nt, mb, K = g.shape
    lam = torch.empty_like(g)
    check(_lib.lib().nns_ode_adjoint_chain_f32(_p(J), _p(g), _p(lam), Nt, mb, K, _stream()), 'nns_ode_adjoint_chain_f32')
    return lam


def basis_expand(coeff, basis):
    """coeff [T, K, C], basis [K, C, P] -> pred [T, C, P]"""
    _f32(coeff, basis)
    T, K, C = coeff.shape
    P = basis.shape[2]
    pred = torch.empty(T, C, P, dtype=torch.float32, device=coeff.device)
    check(_lib.lib().nns_basis_expand_f32(_p(coeff), _p(basis), _p(pred), T, K, C, P, _stream()), 'nns_basis_expand_f32')
    return pred


def basis_expand_bwd(coeff, basis, grad_pred):
    _f32(coeff, basis, grad_pred)
    T, K, C = coeff.shape
    P = basis.shape[2]
    gc, gb = torch.empty_like(coeff), torch.empty_like(basis)
    check(_lib.lib().nns_basis_expand_bwd_f32(_p(coeff), _p(basis), _p(grad_pred), _p(gc), _p(gb), T, K, C, P, _stream()), 'nns_basis_expand_bwd_f32')
    return gc, gb


def basis_loss_fwd(coeff, basis, obs):
    """sum (pred - obs)^2 as a device float64 scalar tensor (pred never materialised)."""
    _f32(coeff, basis, obs)
    T, K, C = coeff.shape
    P = basis.shape[2]
    ss = torch.zeros(1, dtype=torch.float64, device=coeff.device)
    check(_lib.lib().nns_basis_loss_fwd_f32(_p(coeff), _p(basis), _p(obs), _p(ss), T, K, C, P, _stream()), 'nns_basis_loss_fwd_f32')
    return ss


def basis_loss_fused(coeff, basis, obs):
    """ONE sweep over obs: returns (sumsq float64 device scalar, d(sumsq/2)/dcoeff, d(sumsq/2)/dbasis) -- the gradient without the
    upstream / loss factor (nns_basis_loss_fused_f32)."""
    _f32(coeff, basis, obs)
    T, K, C = coeff.shape
    P = basis.shape[2]
    ss = torch.zeros(1, dtype=torch.float64, device=coeff.device)
    gc, gb = torch.empty_like(coeff), torch.empty_like(basis)
    check(_lib.lib().nns_basis_loss_fused_f32(_p(coeff), _p(basis), _p(obs), _p(ss), _p(gc), _p(gb), T, K, C, P, _stream()), 'nns_basis_loss_fused_f32')
    return ss, gc, gb


def basis_loss_bwd(coeff, basis, obs, scale):
    _f32(coeff, basis, obs)
    T, K, C = coeff.shape
    P = basis.shape[2]
    gc, gb = torch.empty_like(coeff), torch.empty_like(basis)
    check(_lib.lib().nns_basis_loss_bwd_f32(_p(coeff), _p(basis), _p(obs), float(scale), _p(gc), _p(gb), T, K, C, P, _stream()), 'nns_basis_loss_bwd_f32')
    return gc, gb


# ----------------------------------------------------------------------------- chorin_spectral
def _f64(*ts):
    for t in ts:
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
            raise TypeError("expected contiguous float64 CUDA/HIP tensors")


def cheb_gemm(A, B, transA=False, transB=False, alpha=1.0, beta=0.0, out=None):
    """Row-major 2-D float64: out = alpha * op(A) @ op(B) + beta * out."""
    _f64(A, B)
    M = A.shape[1] if transA else A.shape[0]
    K = A.shape[0] if transA else A.shape[1]
    N = B.shape[0] if transB else B.shape[1]
    if (B.shape[1] if transB else B.shape[0]) != K:
        raise ValueError("cheb_gemm: inner dimensions differ")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float64, device=A.device)
        beta = 0.0
    _f64(out)
    check(_lib.lib().nns_cheb_gemm_f64(_p(A), A.shape[1], int(transA), _p(B), B.shape[1], int(transB), _p(out), out.shape[1],
                                       M, N, K, float(alpha), float(beta), 1, _stream()), 'nns_cheb_gemm_f64')
    return out


def cheb_helmholtz_rhs(f, un, vn, un1, vn1, fx, fy, f1x, f1y, fxx, fyy, dt):
    _f64(f, un, vn, un1, vn1, fx, fy, f1x, f1y, fxx, fyy)
    F = torch.empty_like(f)
    check(_lib.lib().nns_cheb_helmholtz_rhs_f64(*[_p(t) for t in (f, un, vn, un1, vn1, fx, fy, f1x, f1y, fxx, fyy)], _p(F), f.numel(),
                                                float(dt), _stream()), 'nns_cheb_helmholtz_rhs_f64')
    return F


def cheb_diag_div(Hm, lam_x, lam_y, c0, cx, cy):
    _f64(Hm, lam_x, lam_y)
    out = torch.empty_like(Hm)
    check(_lib.lib().nns_cheb_diag_div_f64(_p(Hm), _p(lam_x), _p(lam_y), _p(out), Hm.shape[0], Hm.shape[1], float(c0), float(cx), float(cy),
                                           _stream()), 'nns_cheb_diag_div_f64')
    return out


def cheb_embed(sol, x0, xN, y0, yN):
    _f64(sol, x0, xN, y0, yN)
    Nx, Ny = sol.shape[0] + 2, sol.shape[1] + 2
    full = torch.empty(Nx, Ny, dtype=torch.float64, device=sol.device)
    check(_lib.lib().nns_cheb_embed_f64(_p(sol), _p(x0), _p(xN), _p(y0), _p(yN), _p(full), Nx, Ny, _stream()), 'nns_cheb_embed_f64')
    return full


# ----------------------------------------------------------------------------- per-pixel MLP (BasisFunc)
def pixel_mlp_fwd(x, weights, biases, bf16=False):
    """x [mb, C_in, nx, ny] (or [mb, C_in, P]); weights: list of [C_out, C_in] (or Conv2d [C_out, C_in, 1, 1]) tensors;
    biases: list of [C_out].  ReLU between layers, none after the last."""
    import ctypes
    _f32(x)
    mb, cin = x.shape[0], x.shape[1]
    P = x[0, 0].numel()
    ws = [w.reshape(w.shape[0], w.shape[1]) for w in weights]
    widths = [cin] + [w.shape[0] for w in ws]
    for i, w in enumerate(ws):
        if w.shape[1] != widths[i]:
            raise ValueError("pixel_mlp_fwd: layer %d expects %d input channels, got %d" % (i, w.shape[1], widths[i]))
    wp = torch.cat([w.reshape(-1) for w in ws]).to(torch.float32).contiguous()
    bp = torch.cat([b.reshape(-1) for b in biases]).to(torch.float32).contiguous()
    y = torch.empty((mb, widths[-1]) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    arr = (ctypes.c_int * len(widths))(*widths)
    check(_lib.lib().nns_pixel_mlp_fwd_f32(_p(x), _p(wp), _p(bp), _p(y), mb, P, arr, len(ws), int(bool(bf16)), _stream()), 'nns_pixel_mlp_fwd_f32')
    return y


def _pixel_mlp_pack(x, weights, biases, what):
    _f32(x)
    mb, cin = x.shape[0], x.shape[1]
    P = x[0, 0].numel()
    ws = [w.reshape(w.shape[0], w.shape[1]) for w in weights]
    widths = [cin] + [w.shape[0] for w in ws]
    for i, w in enumerate(ws):
        if w.shape[1] != widths[i]:
            raise ValueError("%s: layer %d expects %d input channels, got %d" % (what, i, w.shape[1], widths[i]))
    wp = torch.cat([w.reshape(-1) for w in ws]).to(torch.float32).contiguous()
    bp = torch.cat([b.reshape(-1) for b in biases]).to(torch.float32).contiguous()
    return mb, P, ws, widths, wp, bp


def pixel_mlp_bwd(x, gy, weights, biases, bf16=True):
    """Backward of pixel_mlp_fwd: returns (gx like x, [gW_l like weights[l]], [gb_l like biases[l]]).
    bf16=False (float32 operands) supports widths <= 32."""
    import ctypes
    mb, P, ws, widths, wp, bp = _pixel_mlp_pack(x, weights, biases, 'pixel_mlp_bwd')
    _f32(gy)
    if tuple(gy.shape) != (mb, widths[-1]) + tuple(x.shape[2:]):
        raise ValueError("pixel_mlp_bwd: gy has shape %s, expected %s" % (tuple(gy.shape), (mb, widths[-1]) + tuple(x.shape[2:])))
    arr = (ctypes.c_int * len(widths))(*widths)
    nbytes = ctypes.c_size_t(0)
    check(_lib.lib().nns_pixel_mlp_bwd_workspace(arr, len(ws), ctypes.byref(nbytes)), 'nns_pixel_mlp_bwd_workspace')
    work = torch.empty(max(nbytes.value // 4, 1), dtype=torch.float32, device=x.device)
    gx = torch.empty_like(x)
    gW = torch.empty_like(wp)
    gB = torch.empty_like(bp)
    check(_lib.lib().nns_pixel_mlp_bwd_f32(_p(x), _p(gy), _p(wp), _p(bp), _p(gx), _p(gW), _p(gB), mb, P, arr, len(ws), int(bool(bf16)),
                                           _p(work), nbytes.value, _stream()), 'nns_pixel_mlp_bwd_f32')
    gws, gbs, wo, bo = [], [], 0, 0
    for w, b in zip(weights, biases):
        gws.append(gW[wo:wo + w.numel()].reshape(w.shape)); wo += w.numel()
        gbs.append(gB[bo:bo + b.numel()].reshape(b.shape)); bo += b.numel()
    return gx, gws, gbs


class PixelMlpFn(torch.autograd.Function):
    """pixel_mlp_fwd / pixel_mlp_bwd as one autograd node (bf16 or float32 operands, float32 accumulation)."""

    @staticmethod
    def forward(ctx, x, nlayers, bf16, *params):
        weights, biases = params[:nlayers], params[nlayers:]
        ctx.nlayers, ctx.bf16 = nlayers, bool(bf16)
        ctx.save_for_backward(x, *params)
        return pixel_mlp_fwd(x, weights, biases, bf16=bf16)

    @staticmethod
    def backward(ctx, gy):
        x, *params = ctx.saved_tensors
        weights, biases = params[:ctx.nlayers], params[ctx.nlayers:]
        gx, gws, gbs = pixel_mlp_bwd(x.contiguous(), gy.contiguous(), weights, biases, bf16=ctx.bf16)
        return (gx, None, None) + tuple(gws) + tuple(gbs)


# ----------------------------------------------------------------------------- standalone spectral operators
def spec_derivs(f, Lx, Ly, want=('x', 'y', 'lap'), precise=True):
    """Spectral f_x, f_y, lap f of one real float32 field [B, nx, ny] (or [nx, ny]); returns a dict."""
    _f32(f)
    B, nx, ny = _dims(f)
    out = {k: torch.empty_like(f) for k in want}
    ptr = lambda k: _p(out[k]) if k in out else None
    check(_lib.lib().nns_spec_derivs_f32(_p(f), ptr('x'), ptr('y'), ptr('lap'), B, nx, ny, float(Lx), float(Ly), _prec(precise), _stream()),
          'nns_spec_derivs_f32')
    return out


def spec_residual_bwd(u, v, g_u, g_v, g_div, dt, Lx, Ly, rho, nu, precise=True, want_prev=True):
    """Vector-Jacobian product of spec_residual (oracle/periodic.py: spectral_residual_vjp) by the fused two-pass
    backward kernels (nns_spec_residual_bwd_f32): returns (grad_u, grad_v, grad_p, grad_u_prev, grad_v_prev)."""
    suf, (B, nx, ny) = _chk(u, v, g_u, g_v, g_div)
    if suf != '_f32':
        raise TypeError("spec_residual_bwd: float32 fields")
    gu, gv, gp = torch.empty_like(u), torch.empty_like(u), torch.empty_like(u)
    gup, gvp = (torch.empty_like(u), torch.empty_like(u)) if want_prev else (None, None)
    check(_lib.lib().nns_spec_residual_bwd_f32(_p(u), _p(v), _p(g_u), _p(g_v), _p(g_div), _p(gu), _p(gv), _p(gp),
                                               _p(gup) if want_prev else None, _p(gvp) if want_prev else None,
                                               B, nx, ny, dt, Lx, Ly, rho, nu, _prec(precise), _stream()), 'nns_spec_residual_bwd_f32')
    return gu, gv, gp, gup, gvp


def spec_residual_bwd_composed(u, v, g_u, g_v, g_div, dt, Lx, Ly, rho, nu, precise=True):
    """The same product assembled from the standalone spectral-derivative kernel (8 launches) and elementwise tensor
    ops: kept as an independent cross-check of the fused kernels (tests), not used on the training path."""
    _f32(u, v, g_u, g_v, g_div)
    d = lambda f, *want: spec_derivs(f.contiguous(), Lx, Ly, want, precise)
    du, dv = d(u, 'x', 'y'), d(v, 'x', 'y')
    da, db = d(g_u, 'x', 'lap'), d(g_v, 'y', 'lap')
    grad_u = g_u / dt + g_u * du['x'] + g_v * dv['x'] - d(g_u * u + g_div, 'x')['x'] - d(g_u * v, 'y')['y'] - nu * da['lap']
    grad_v = g_v / dt + g_u * du['y'] + g_v * dv['y'] - d(g_v * u, 'x')['x'] - d(g_v * v + g_div, 'y')['y'] - nu * db['lap']
    grad_p = -(da['x'] + db['y']) / rho
    return grad_u, grad_v, grad_p


class SpecResidualFn(torch.autograd.Function):
    """spec_residual as an autograd node."""

    @staticmethod
    def forward(ctx, u, v, p, u_prev, v_prev, dt, Lx, Ly, rho, nu, precise):
        u, v, p, u_prev, v_prev = (t.contiguous() for t in (u, v, p, u_prev, v_prev))
        ctx.save_for_backward(u, v)
        ctx.consts = (dt, Lx, Ly, rho, nu, precise)
        return spec_residual(u, v, p, u_prev, v_prev, dt, Lx, Ly, rho, nu, precise)

    @staticmethod
    def backward(ctx, g_u, g_v, g_d):
        u, v = ctx.saved_tensors
        dt, Lx, Ly, rho, nu, precise = ctx.consts
        zero = lambda g: torch.zeros_like(u) if g is None else g.contiguous()
        want_prev = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        return spec_residual_bwd(u, v, zero(g_u), zero(g_v), zero(g_d), dt, Lx, Ly, rho, nu, precise, want_prev) + (None,) * 6


def spec_rfft2(f):
    """numpy.fft.rfft2 of float32 [B, nx, ny] -> complex64 [B, nx, ny//2+1]."""
    _f32(f)
    B, nx, ny = _dims(f)
    spec = torch.empty((B, nx, ny // 2 + 1, 2), dtype=torch.float32, device=f.device)
    check(_lib.lib().nns_spec_rfft2_f32(_p(f), _p(spec), B, nx, ny, _stream()), 'nns_spec_rfft2_f32')
    return torch.view_as_complex(spec)


def spec_irfft2(spec, ny):
    """numpy.fft.irfft2(spec, s=(nx, ny)) for complex64 [B, nx, ny//2+1]; the input is not modified (a scratch copy is)."""
    s = torch.view_as_real(spec.contiguous()).clone()
    _f32(s)
    B, nx = spec.shape[0], spec.shape[1]
    f = torch.empty((B, nx, ny), dtype=torch.float32, device=spec.device)
    check(_lib.lib().nns_spec_irfft2_f32(_p(s), _p(f), B, nx, ny, _stream()), 'nns_spec_irfft2_f32')
    return f


# ----------------------------------------------------------------------------- physics-informed loss head
_PINN_WS = {}


def _pinn_ws(device):
    """The head's workspace (per-block partial sums + the arrival counter), zeroed once per device and stream."""
    key = (device, _stream())
    if key not in _PINN_WS:
        _PINN_WS[key] = torch.zeros(int(_lib.lib().nns_pinn_workspace_bytes()) // 8 + 1, dtype=torch.float64, device=device)
    return _PINN_WS[key]


class PinnHeadFn(torch.autograd.Function):
    """total, data, phys = head(mlp_out, state, target): pred = state + mlp_out, data = mean (pred - target)^2, phys = mean-square residual of pred,
    total = data + lam phys (nns/neural_spectral/physics_informed.py) as ONE autograd node: nns_pinn_assemble_f32 -> the residual kernel ->
    nns_pinn_loss_f32 forward; the residual adjoint -> nns_pinn_combine_f32 backward.  mlp_out, state, target: contiguous float32 [batch, 3, ...]
    (channel-major fields: batch = 1); `residual` = (kind, consts): ('fd', (dt, dx, dy, rho, nu, stencil)) or ('spectral', (dt, Lx, Ly, rho, nu, precise));
    dims = the fields' (B, nx, ny)."""

    @staticmethod
    def forward(ctx, out, state, target, residual, dims, lam, w_div):
        for t in (out, state) + ((target,) if target is not None else ()):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape == out.shape):
                raise ValueError("PinnHeadFn: mlp_out, state and target must be contiguous float32 device tensors of one shape")
        B, nx, ny = dims
        batch, npix = out.shape[0], out[0, 0].numel()
        if out.shape[1] != 3 or batch * npix != B * nx * ny:
            raise ValueError("PinnHeadFn: fields [batch, 3, ...] with batch x pixels = %d x %d x %d expected, got %s" % (B, nx, ny, tuple(out.shape)))
        L, st, ws = _lib.lib(), _stream(), _pinn_ws(out.device)
        new = lambda: torch.empty((B, nx, ny), dtype=torch.float32, device=out.device)
        u, v, p = new(), new(), new()
        if batch == 1:                                   # channel-major: the state's channels ARE contiguous fields
            u_prev, v_prev, pu, pv = state[0, 0].reshape(B, nx, ny), state[0, 1].reshape(B, nx, ny), None, None
        else:
            u_prev, v_prev = new(), new()
            pu, pv = _p(u_prev), _p(v_prev)
        check(L.nns_pinn_assemble_f32(_p(out), _p(state), _p(target) if target is not None else None, _p(u), _p(v), _p(p), pu, pv, _p(ws),
                                      batch, npix, st), 'nns_pinn_assemble_f32')
        kind, c = residual
        if kind == 'fd':
            r = fd_residual(u, v, p, u_prev, v_prev, *c)
        else:
            r = spec_residual(u, v, p, u_prev, v_prev, *c)
        out3 = torch.empty(3, dtype=torch.float32, device=out.device)
        n = B * nx * ny
        n_data = 3.0 * n if target is not None else 0.0
        check(L.nns_pinn_loss_f32(_p(r[0]), _p(r[1]), _p(r[2]), n, _p(ws), n_data, float(lam), float(w_div), _p(out3), st), 'nns_pinn_loss_f32')
        ctx.save_for_backward(u, v, p, r[0], r[1], r[2], target)
        ctx.consts = (residual, (B, nx, ny), batch, npix, float(lam), n, n_data, out.shape)
        ctx.set_materialize_grads(False)
        return out3.unbind(0)                            # total, data, phys

    @staticmethod
    def backward(ctx, g_total, g_data, g_phys):
        u, v, p, ru, rv, rd, target = ctx.saved_tensors
        residual, (B, nx, ny), batch, npix, lam, n, n_data, shape = ctx.consts
        kind, c = residual
        if kind == 'fd':
            gu, gv, gp, _, _ = fd_residual_bwd(u, v, ru, rv, rd, *c, want_prev=False)
        else:
            gu, gv, gp, _, _ = spec_residual_bwd(u, v, ru, rv, rd, *c, want_prev=False)
        zero = lambda g: torch.zeros((), dtype=torch.float32, device=u.device) if g is None else g.to(torch.float32)
        if g_data is None and g_phys is None:            # the training step: total.backward()
            up_d = up_p = zero(g_total).contiguous()
            c_phys = 2.0 * lam / n
        else:                                            # someone differentiates the parts: d/d data = g_total + g_data, d/d phys = lam g_total + g_phys
            up_d = (zero(g_total) + zero(g_data)).contiguous()
            up_p = (lam * zero(g_total) + zero(g_phys)).contiguous()
            c_phys = 2.0 / n
        grad = torch.empty(shape, dtype=torch.float32, device=u.device)
        check(_lib.lib().nns_pinn_combine_f32(_p(gu), _p(gv), _p(gp), _p(u), _p(v), _p(p), _p(target) if target is not None else None,
                                              _p(up_d), _p(up_p), 2.0 / n_data if n_data else 0.0, c_phys, _p(grad), batch, npix, _stream()),
              'nns_pinn_combine_f32')
        return grad, None, None, None, None, None, None
