"""The GRU variant of the spectral model (reference: src/neural_spectral/spectral_rnn.py):
the coefficient dynamics w_k(t) come from a GRU instead of an ODE, the field is the same basis expansion
u(x, y, t) = sum_k w_k(t) f_k(x, y).  The recurrence runs on torch's GRU (MIOpen on ROCm); the expansion, the training
loss and their gradients run on the fused HIP kernels shared with ``spectral_ode.PDEFunc`` (nns_basis_expand_f32,
nns_basis_loss_fwd/bwd_f32).  Same class surface and state-dict names as the reference (:13-79)."""
import torch
import torch.nn as nn

from .spectral_ode import expand, _BasisLossFn, _require_device, AverageMeter      # noqa: F401  (re-exported like the reference)


class PDEFunc(nn.Module):
    def __init__(self, K, nx, ny):
        super().__init__()
        self.K = K
        self.nx, self.ny = nx, ny
        self.init_coeffs = nn.Parameter(torch.normal(torch.zeros(self.K * 3), 1))
        self.basis_coeffs = nn.GRU(self.K * 3, self.K * 3, batch_first=True)
        self.basis_fns = nn.ParameterList([
            nn.Parameter(torch.normal(torch.zeros(3, self.nx, self.ny), 1))
            for _ in range(self.K)
        ])

    def rnnint(self, init_coeff, nt):
        """The GRU fed its own output for nt steps from init_coeff [mb, K*3] (:35-43); returns the steps stacked as [nt * mb, K*3]."""
        x, h, steps = init_coeff[:, None, :], None, []
        for _ in range(nt):
            x, h = self.basis_coeffs(x, h)
            steps.append(x[:, 0])
        return torch.cat(steps)

    def _coeff(self, mb, nt):
        return self.rnnint(self.init_coeffs.unsqueeze(0).repeat(mb, 1), nt).view(nt * mb, self.K, 3)      # (:52) view(nt, mb, K, 3)

    def _basis(self):
        return torch.stack([f for f in self.basis_fns]).reshape(self.K, 3, self.nx * self.ny)

    def forward(self, grid0, t):
        mb, nt = grid0.size(0), t.size(0)
        return expand(self._coeff(mb, nt), self._basis()).view(nt, mb, 3, self.nx, self.ny)

    def loss(self, grid0, t, obs):
        """Fused ``torch.norm(self(grid0, t) - obs, p=2)`` (the reference's training objective)."""
        mb, nt = grid0.size(0), t.size(0)
        coeff, basis = self._coeff(mb, nt), self._basis()
        o = obs.reshape(nt * mb, 3, self.nx * self.ny)
        _require_device('basis loss', coeff, basis, o)
        return _BasisLossFn.apply(coeff, basis, o)

    def basis_weight_mat(self):
        return torch.stack([f.reshape(-1) for f in self.basis_fns])

    def diversity_penalty(self):
        return 1. / torch.pdist(self.basis_weight_mat(), p=2).sum()
