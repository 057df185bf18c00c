"""Mirror of the reference's ``src/neural_spectral/spectral_ode.py``: ``ODEFunc``, ``PDEFunc``, ``BasisFunc``,
``AverageMeter`` with identical constructor arguments, parameter / state-dict names
(``init_coeffs``, ``basis_coeffs.net.{0,2,4}.{weight,bias}``, ``basis_fns.{k}``), initialisation
(weights ~ N(0, 0.1), biases 0, :28-31; coefficients and bases ~ N(0, 1), :53,:58) and forward semantics.

u(x, y, t) = sum_k w_k(t) f_k(x, y)  (:62-81): the coefficient ODE runs as one fused RK4-MLP kernel, the
expansion as one kernel; ``PDEFunc.loss`` is the fused training path (prediction never materialised).
"""
import torch
import torch.nn as nn

from .. import ops
from .anode import odesolver_adjoint as odesolver


class ODEFunc(nn.Module):
    """Model basis coefficients as an ODE wrt time (spectral_ode.py:14-34)."""

    def __init__(self, K):
        super().__init__()
        self.K = K
        self.net = nn.Sequential(
            nn.Linear(self.K, 128),
            nn.ReLU(inplace=True),
            nn.Linear(128, 128),
            nn.ELU(inplace=True),
            nn.Linear(128, self.K),
        )
        for m in self.net.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, mean=0, std=0.1)
                nn.init.constant_(m.bias, val=0)

    def forward(self, t, coeff):
        return self.net(coeff)


class _BasisExpandFn(torch.autograd.Function):
    """pred[T, C, P] = sum_k coeff[T, K, C] * basis[K, C, P]"""

    @staticmethod
    def forward(ctx, coeff, basis):
        coeff, basis = coeff.contiguous(), basis.contiguous()
        ctx.save_for_backward(coeff, basis)
        return ops.basis_expand(coeff, basis)

    @staticmethod
    def backward(ctx, g):
        coeff, basis = ctx.saved_tensors
        return ops.basis_expand_bwd(coeff, basis, g.contiguous())


class _BasisSumsqFn(torch.autograd.Function):
    """sum (sum_k coeff basis - obs)^2 as a float64 device scalar, without materialising the prediction.  When a gradient
    will be wanted the forward sweep also accumulates d(sumsq / 2) / d(coeff, basis) (nns_basis_loss_fused_f32): the
    observations -- the only large stream -- cross HBM ONCE per training step, and the backward is two small scalings on the
    device (no host read of the loss)."""

    @staticmethod
    def forward(ctx, coeff, basis, obs):
        coeff, basis, obs = coeff.contiguous(), basis.contiguous(), obs.contiguous()
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            ss, gc, gb = ops.basis_loss_fused(coeff, basis, obs)
            ctx.save_for_backward(gc, gb)
        else:
            ss = ops.basis_loss_fwd(coeff, basis, obs)
        return ss.reshape(())

    @staticmethod
    def backward(ctx, g):
        gc, gb = ctx.saved_tensors
        s = (2.0 * g).to(torch.float32)                       # d sumsq = 2 * d(sumsq / 2)
        return gc * s, gb * s, None


def basis_sumsq(coeff, basis, obs):
    return _BasisSumsqFn.apply(coeff, basis, obs)


def basis_norm_loss(coeff, basis, obs):
    """|| sum_k coeff basis - obs ||_2 (spectral_ode.py:182), float32 scalar."""
    return torch.sqrt(_BasisSumsqFn.apply(coeff, basis, obs)).to(torch.float32)


class _BasisLossFn(object):
    """Kept name of the fused training objective (used by spectral_ode2 / spectral_rnn): see basis_norm_loss."""
    apply = staticmethod(basis_norm_loss)


def _require_device(what, *ts):
    """The expansion and its loss exist as HIP kernels only: no silent CPU / eager fallback."""
    for t in ts:
        if not (t.is_cuda and t.dtype == torch.float32):
            raise RuntimeError("%s needs float32 tensors on the HIP device (got %s on %s): the product has no CPU path"
                               % (what, t.dtype, t.device))


def expand(coeff, basis):
    """coeff [T, K, C], basis [K, C, P] -> [T, C, P] by the fused HIP kernel (K <= 32; larger K fails in the C ABI)."""
    _require_device('basis expansion', coeff, basis)
    return _BasisExpandFn.apply(coeff, basis)


class PDEFunc(nn.Module):
    """u(x,y,t) = sum_k w_k(t) f_k(x,y)  (spectral_ode.py:37-97)."""

    def __init__(self, K, nx, ny):
        super().__init__()
        self.K = K
        self.nx, self.ny = nx, ny
        self.init_coeffs = nn.Parameter(torch.normal(torch.zeros(self.K * 3), 1))
        self.basis_coeffs = ODEFunc(self.K * 3)
        self.basis_fns = nn.ParameterList([
            nn.Parameter(torch.normal(torch.zeros(3, self.nx, self.ny), 1))
            for _ in range(self.K)
        ])

    def _coeff(self, mb, nt):
        # The reference integrates mb COPIES of the one init_coeffs (:69 repeat, :70): every member of the batch follows the same
        # trajectory.  It is integrated once and broadcast; the backward of the broadcast sums the members' gradients, so the ODE
        # adjoint runs on ONE row (time-parallel, anode._OdeMlpFn) instead of mb identical ones.
        coeff = odesolver(self.basis_coeffs, self.init_coeffs.unsqueeze(0), {'Nt': nt, 'method': 'RK4'})      # [nt, 1, 3K]
        return coeff.expand(nt, mb, 3 * self.K).reshape(nt * mb, self.K, 3)                                     # (:71) view(nt, mb, K, 3)

    def _basis(self):
        return torch.stack([f for f in self.basis_fns]).reshape(self.K, 3, self.nx * self.ny)

    def forward(self, grid0, t):
        # grid0 = mb x 3 x nx x ny (only its batch size is used, :67), t = nt (only its length is used)
        mb, nt = grid0.size(0), t.size(0)
        soln = expand(self._coeff(mb, nt), self._basis())
        return soln.view(nt, mb, 3, self.nx, self.ny)

    def loss(self, grid0, t, obs):
        """Fused ``torch.norm(self(grid0, t) - obs, p=2)`` (the training objective, :181-182)."""
        mb, nt = grid0.size(0), t.size(0)
        coeff, basis = self._coeff(mb, nt), self._basis()
        o = obs.reshape(nt * mb, 3, self.nx * self.ny)
        _require_device('basis loss', coeff, basis, o)
        return basis_norm_loss(coeff, basis, o)

    def sumsq(self, grid0, t, obs):
        """The SQUARE of `loss` as a float64 scalar: additive over shards of the ensemble, which is what data-parallel
        training all-reduces (nns.data_parallel.norm_loss_step)."""
        mb, nt = grid0.size(0), t.size(0)
        coeff, basis = self._coeff(mb, nt), self._basis()
        o = obs.reshape(nt * mb, 3, self.nx * self.ny)
        _require_device('basis loss', coeff, basis, o)
        return basis_sumsq(coeff, basis, o)

    def basis_weight_mat(self):
        """[K, 3 nx ny]: one flattened basis function per row (spectral_ode.py:83-88)."""
        return torch.stack([f.reshape(-1) for f in self.basis_fns])

    def diversity_penalty(self):
        """1 / sum of pairwise L2 distances between the basis functions (spectral_ode.py:90-97; logging only, :184-186).  The
        reference's double loop also visits i == j, whose distance is zero: the sum over pairs i < j is the same number."""
        return 1. / torch.pdist(self.basis_weight_mat(), p=2).sum()


class BasisFunc(nn.Module):
    """A basis to build up a function (spectral_ode.py:100-119): per-pixel MLP as 1x1 convolutions."""

    def __init__(self, nx, ny):
        super().__init__()
        self.nx, self.ny = nx, ny
        self.net = nn.Sequential(
            nn.Conv2d(3, 16, 1),
            nn.ReLU(inplace=True),
            nn.Conv2d(16, 32, 1),
            nn.ReLU(inplace=True),
            nn.Conv2d(32, 32, 1),
            nn.ReLU(inplace=True),
            nn.Conv2d(32, 16, 1),
            nn.ReLU(inplace=True),
            nn.Conv2d(16, 3, 1),
        )

    def forward(self, grid):
        return self.net(grid)

    def fused_forward(self, grid, bf16=False):
        """Inference through ONE fused HIP kernel (all five 1x1 convolutions chained in MFMA accumulators,
        csrc/pixel_mlp_kernels.hip) instead of five convolution launches; no autograd graph is recorded."""
        convs = [m for m in self.net if isinstance(m, nn.Conv2d)]
        with torch.no_grad():
            return ops.pixel_mlp_fwd(grid.contiguous(), [c.weight for c in convs], [c.bias for c in convs], bf16=bf16)


class PixelMLP(nn.Module):
    """The configurable-depth generalisation of BasisFunc used by BASELINE configs 2 and 3 (e.g. depth 4 width 32
    float32; depth 8 width 64 bfloat16): per-pixel Linear-ReLU stack, 3 -> width -> ... -> 3."""

    def __init__(self, depth=4, width=32, c_in=3, c_out=3):
        super().__init__()
        dims = [c_in] + [width] * (depth - 1) + [c_out]
        self.weights = nn.ParameterList([nn.Parameter(torch.randn(dims[i + 1], dims[i]) / dims[i] ** 0.5) for i in range(depth)])
        self.biases = nn.ParameterList([nn.Parameter(torch.zeros(dims[i + 1])) for i in range(depth)])

    def forward(self, grid, bf16=False):
        """Inference (no autograd graph); `train_forward` is the differentiable bf16 path."""
        with torch.no_grad():
            return ops.pixel_mlp_fwd(grid.contiguous(), list(self.weights), list(self.biases), bf16=bf16)

    def train_forward(self, grid, bf16=True):
        """Forward recorded as ONE autograd node whose backward is the fused HIP kernel (nns_pixel_mlp_bwd_f32: forward
        recomputed in registers, no saved activations).  bf16=False: float32 operands, widths <= 32."""
        return ops.PixelMlpFn.apply(grid.contiguous(), len(self.weights), bool(bf16), *self.weights, *self.biases)


class AverageMeter(object):
    """Running weighted mean of a logged scalar, with the attribute names the training loop reads (`val`, `sum`, `count`,
    `avg`; spectral_ode.py:122-137)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.sum = self.count = self.avg = 0

    def update(self, val, n=1):
        self.val, self.sum, self.count = val, self.sum + val * n, self.count + n
        self.avg = self.sum / self.count
