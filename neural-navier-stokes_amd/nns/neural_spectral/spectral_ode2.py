"""Mirror of the reference's ``src/neural_spectral/spectral_ode2.py``: one ODEFunc(K) and K scalar-field bases
per channel (u, v, p); parameter names ``{u,v,p}_init_coeffs``, ``{u,v,p}_basis_coeffs.net.*``,
``{u,v,p}_basis_fns.{k}`` (:49-67)."""
import torch
import torch.nn as nn

from .anode import odesolver_adjoint as odesolver
from .spectral_ode import ODEFunc, AverageMeter, expand, _BasisLossFn, _require_device  # noqa: F401


class PDEFunc(nn.Module):
    def __init__(self, K, nx, ny):
        super().__init__()
        self.K = K
        self.nx, self.ny = nx, ny
        self.u_init_coeffs = nn.Parameter(torch.normal(torch.zeros(self.K), 1))
        self.v_init_coeffs = nn.Parameter(torch.normal(torch.zeros(self.K), 1))
        self.p_init_coeffs = nn.Parameter(torch.normal(torch.zeros(self.K), 1))
        self.u_basis_coeffs = ODEFunc(self.K)
        self.v_basis_coeffs = ODEFunc(self.K)
        self.p_basis_coeffs = ODEFunc(self.K)
        mk = lambda: nn.ParameterList([nn.Parameter(torch.normal(torch.zeros(self.nx, self.ny), 1)) for _ in range(self.K)])
        self.u_basis_fns = mk()
        self.v_basis_fns = mk()
        self.p_basis_fns = mk()

    def _coeff(self, mb, nt):
        cs = []
        for init, f in ((self.u_init_coeffs, self.u_basis_coeffs), (self.v_init_coeffs, self.v_basis_coeffs),
                        (self.p_init_coeffs, self.p_basis_coeffs)):
            c = odesolver(f, init.unsqueeze(0), {'Nt': nt, 'method': 'RK4'})                     # [nt, 1, K]: mb identical copies (:75-83)
            cs.append(c.expand(nt, mb, self.K).reshape(nt * mb, self.K))
        return torch.stack(cs, dim=2)                                                             # [T, K, 3]

    def _basis(self):
        per = [torch.stack([f for f in fl]) for fl in (self.u_basis_fns, self.v_basis_fns, self.p_basis_fns)]   # 3 x [K, nx, ny]
        return torch.stack(per, dim=1).reshape(self.K, 3, self.nx * self.ny)

    def forward(self, grid0, t):
        mb, nt = grid0.size(0), t.size(0)
        soln = expand(self._coeff(mb, nt), self._basis())
        return soln.view(nt, mb, 3, self.nx, self.ny)

    def loss(self, grid0, t, obs):
        mb, nt = grid0.size(0), t.size(0)
        coeff, basis = self._coeff(mb, nt), self._basis()
        o = obs.reshape(nt * mb, 3, self.nx * self.ny)
        _require_device('basis loss', coeff, basis, o)
        return _BasisLossFn.apply(coeff, basis, o)
