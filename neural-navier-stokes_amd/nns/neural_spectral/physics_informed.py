"""Physics-informed training step: a field-prediction MLP advances (u, v, p) by one
time step and is trained on  data loss + lambda * Navier-Stokes residual of its own prediction  -- the hypothesis the
reference states but never implements (src/neural_spectral/derivations/derivation.tex:25-34, "Neural Residual PDEs").

Everything on the gradient path is a HIP kernel behind the C ABI:
    PixelMLP.train_forward      nns_pixel_mlp_fwd_f32 / nns_pixel_mlp_bwd_f32   (bf16 MFMA, layers chained in registers)
    ResidualEngine.differentiable   nns_fd_residual_* / nns_spec_residual_* and their adjoints
torch supplies the autograd tape, the elementwise loss reductions and the optimiser."""
import torch

from .spectral_ode import PixelMLP
from ..periodic import ResidualEngine


class FieldStepper(torch.nn.Module):
    """(u, v, p)_t -> (u, v, p)_{t+1} = (u, v, p)_t + MLP((u, v, p)_t) per pixel (residual connection: the identity is a
    good first guess for a small time step)."""

    def __init__(self, depth=8, width=64):
        super().__init__()
        self.mlp = PixelMLP(depth, width)
        with torch.no_grad():
            self.mlp.weights[-1].mul_(0.1)

    def forward(self, state):
        return state + self.mlp.train_forward(state)

    def forward_cm(self, state_cm):
        """Channel-major fields [3, B, nx, ny]: the per-pixel MLP sees them as ONE batch item of B nx ny pixels, so the
        prediction's channels come out as three contiguous [B, nx, ny] fields -- no per-channel copies around the residual."""
        c, B, nx, ny = state_cm.shape
        out = self.mlp.train_forward(state_cm.reshape(1, c, B * nx * ny))
        return state_cm + out.reshape(c, B, nx, ny)


def physics_informed_loss(model, engine, state, target=None, lam=1.0, w_div=1.0, layout='bchw'):
    """state, target: float32 fields (channels u, v, p), [B, 3, nx, ny] (layout='bchw') or channel-major [3, B, nx, ny]
    (layout='cm': the residual kernels take the prediction's channels in place, and their gradients return by one
    stack instead of three strided scatters).  Returns (total, data, physics)."""
    if layout == 'cm':
        pred = model.forward_cm(state)
        u, v, p = torch.unbind(pred, 0)                                   # contiguous views; backward = one stack
        phys = engine.physics_loss(u, v, p, state[0], state[1], w_div=w_div)
    else:
        pred = model(state)
        u, v, p = (pred[:, c].contiguous() for c in range(3))
        phys = engine.physics_loss(u, v, p, state[:, 0].contiguous(), state[:, 1].contiguous(), w_div=w_div)
    data = ((pred - target) ** 2).mean() if target is not None else torch.zeros((), device=pred.device)
    return data + lam * phys, data, phys


def train_step(model, engine, optimizer, state, target=None, lam=1.0, w_div=1.0, bucket=None, layout='bchw'):
    """One optimiser step.  Data-parallel runs pass `bucket` (nns.data_parallel.FlatGradAllReduce over the model's
    parameters): each rank works on its shard of the batch and the gradients are averaged with ONE all-reduce."""
    if bucket is not None:
        bucket.zero_()
    else:
        optimizer.zero_grad(set_to_none=True)
    total, data, phys = physics_informed_loss(model, engine, state, target, lam, w_div, layout=layout)
    total.backward()
    if bucket is not None:
        bucket.reduce_()
    optimizer.step()
    return total.detach(), data.detach(), phys.detach()
