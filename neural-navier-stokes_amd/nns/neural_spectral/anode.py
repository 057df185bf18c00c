"""ANODE fixed-step ODE solvers: mirror of ``src/neural_spectral/anode`` of the reference.

  odesolver(func, z0, options)          anode/odesolver.py:21-37   options = {'Nt': int, 'method': 'Euler'|'RK2'|'RK4'}
  odesolver_adjoint(func, z0, options)  anode/adjoint.py:73-76     same values; gradients by recomputation

Semantics kept: dt = 1/Nt, the result stacks y_1 .. y_Nt (y_0 excluded, time_stepper.py:35-45); an
unsupported method prints 'error unsupported method passed' and returns None (odesolver.py:32-34).

When ``func`` is an ``ODEFunc`` (Linear-ReLU-Linear-ELU-Linear, hidden 128, K <= 32) on a HIP device in
float32, the whole integration is ONE persistent kernel launch (``nns_ode_mlp_fwd_f32``) and its backward
ONE launch (``nns_ode_mlp_bwd_f32``: recomputes the stages from the stored states exactly as ANODE's
checkpointing adjoint does, anode/adjoint.py:52-70).  Any other callable falls back to the generic
stepper below (plain torch ops on whatever device the caller uses) -- the reference's own behaviour.
"""
import torch

from .. import ops

_METHODS = ('Euler', 'RK2', 'RK4')
_parallel_rows = None


def _parallel_row_limit():
    """The time-parallel adjoint is used while its Nt * mb * K single-step rows fill the chip about twice over (16 rows per workgroup, one
    workgroup per CU); beyond that the sequential kernel, whose cost does not grow with K, does less work."""
    global _parallel_rows
    if _parallel_rows is None:
        from .. import _lib
        _parallel_rows = 2 * _lib.device_info()['cu_count'] * 16
    return _parallel_rows


class _OdeMlpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z0, W0, b0, W1, b1, W2, b2, Nt, method):
        z0c = z0.contiguous()
        ps = [t.detach().contiguous() for t in (W0, b0, W1, b1, W2, b2)]
        out = ops.ode_mlp_fwd(z0c.detach(), *ps, Nt, method)
        ctx.save_for_backward(z0c.detach(), *ps, out)
        ctx.Nt, ctx.method = Nt, method
        return out

    @staticmethod
    def backward(ctx, grad_out):
        z0, W0, b0, W1, b1, W2, b2, out = ctx.saved_tensors
        Nt, method = ctx.Nt, ctx.method
        mb, K = z0.shape
        grad_out = grad_out.contiguous()
        if Nt > 1 and mb * Nt * K <= _parallel_row_limit():
            # TIME-PARALLEL adjoint.  The checkpointing adjoint (anode/adjoint.py:52-70) walks the Nt steps backwards one after the
            # other -- one workgroup's worth of work per 16 rows, Nt dependent steps deep (66 us each on one CU).  But the
            # Jacobian of step s depends only on its stored input state y_s, so: (1) all Nt * mb step Jacobians at once, as the
            # backward of Nt * mb * K independent single steps with unit upstream vectors; (2) the adjoint recurrence itself on
            # those K x K matrices (Nt tiny mat-vecs); (3) the parameter gradients as the backward of Nt * mb independent single
            # steps with the adjoints of (2).  Same numbers to float32 summation order.
            p = (W0, b0, W1, b1, W2, b2)
            ys = torch.cat([z0[None], out[:-1]])                                              # y_s, the input state of step s: [Nt, mb, K]
            rows = ys[:, :, None, :].expand(Nt, mb, K, K).reshape(Nt * mb * K, K).contiguous()
            eye = torch.eye(K, dtype=z0.dtype, device=z0.device).repeat(Nt * mb, 1)
            J, _ = ops.ode_mlp_bwd_steps(rows, *p, eye, 1. / Nt, method, want_param_grads=False)   # J[s, b, i, :] = (dy_{s+1} / dy_s)^T e_i
            lam = ops.ode_adjoint_chain(J.view(Nt, mb, K, K), grad_out)
            gy, gs = ops.ode_mlp_bwd_steps(ys.reshape(Nt * mb, K).contiguous(), *p, lam.reshape(Nt * mb, K), 1. / Nt, method)
            return (gy.view(Nt, mb, K)[0].contiguous(), *gs, None, None)
        gz0, gs = ops.ode_mlp_bwd(z0, W0, b0, W1, b1, W2, b2, out, grad_out, Nt, method)
        return (gz0, *gs, None, None)


def _fusable(func, z0):
    from .spectral_ode import ODEFunc
    if not isinstance(func, ODEFunc) or not isinstance(z0, torch.Tensor) or not z0.is_cuda or z0.dtype != torch.float32:
        return False
    lin = [func.net[0], func.net[2], func.net[4]]
    return (z0.dim() == 2 and lin[0].out_features == 128 and lin[1].in_features == 128 and lin[1].out_features == 128
            and lin[2].in_features == 128 and lin[0].in_features == z0.shape[1] <= 32 and lin[2].out_features == z0.shape[1]
            and all(l.weight.is_cuda and l.weight.dtype == torch.float32 and l.bias is not None for l in lin))


def _generic(func, z0, Nt, method):
    """time_stepper.py:35-45 + scheme.py:21-42 for an arbitrary callable func(t, y)."""
    y = z0
    dt = 1. / float(Nt)
    out = []
    for n in range(Nt):
        t = 0 + n * dt
        if method == 'Euler':
            y = y + dt * func(t, y)
        elif method == 'RK2':
            k1 = dt * func(t, y)
            k2 = dt * func(t + dt / 2.0, y + 1.0 / 2.0 * k1)
            y = y + k2
        else:
            k1 = dt * func(t, y)
            k2 = dt * func(t + dt / 2.0, y + 1.0 / 2.0 * k1)
            k3 = dt * func(t + dt / 2.0, y + 1.0 / 2.0 * k2)
            k4 = dt * func(t + dt, y + k3)
            y = y + 1.0 / 6.0 * k1 + 1.0 / 3.0 * k2 + 1.0 / 3.0 * k3 + 1.0 / 6.0 * k4
        out.append(y)
    return torch.stack(out)


def odesolver(func, z0, options=None):
    if options == None:          # noqa: E711  (reference: Nt = 2 but then options['method'] raises TypeError, :22-26)
        Nt = 2
    else:
        Nt = options['Nt']
    method = options['method']
    if method not in _METHODS:
        print('error unsupported method passed')
        return
    if _fusable(func, z0):
        n = func.net
        return _OdeMlpFn.apply(z0, n[0].weight, n[0].bias, n[2].weight, n[2].bias, n[4].weight, n[4].bias, int(Nt), method)
    from .spectral_ode import ODEFunc
    if isinstance(func, ODEFunc):
        # the product's own MLP has HIP kernels only: no silent eager / CPU integration
        raise RuntimeError("odesolver(ODEFunc): needs float32 parameters and a [mb, K <= 32] float32 state on the HIP device "
                           "(got z0 %s %s on %s)" % (tuple(z0.shape), z0.dtype, z0.device))
    return _generic(func, z0, int(Nt), method)          # arbitrary user callable func(t, y): the reference's generic stepper


def odesolver_adjoint(func, z0, options=None):
    """anode/adjoint.py:73-76.  Values equal odesolver's (asserted for the reference in oracle/capture.py); the fused
    backward always recomputes, so for an ODEFunc this IS the checkpointing adjoint; for a generic callable the
    gradients of plain autograd through the stepper are the same numbers."""
    return odesolver(func, z0, options)
