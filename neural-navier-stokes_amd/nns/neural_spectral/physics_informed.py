"""Physics-informed training step: a field-prediction MLP advances (u, v, p) by one
time step and is trained on  data loss + lambda * Navier-Stokes residual of its own prediction  -- the hypothesis the
reference states but never implements (src/neural_spectral/derivations/derivation.tex:25-34, "Neural Residual PDEs").

Everything on the gradient path is a HIP kernel behind the C ABI:
    PixelMLP.train_forward      nns_pixel_mlp_fwd_f32 / nns_pixel_mlp_bwd_f32   (bf16 MFMA, layers chained in registers)
    ResidualEngine.differentiable   nns_fd_residual_* / nns_spec_residual_* and their adjoints
    ops.PinnHeadFn                  nns_pinn_assemble_f32 / nns_pinn_loss_f32 / nns_pinn_combine_f32 (round 4: the loss head)
torch supplies the autograd tape and the optimiser."""
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


def physics_informed_loss(model, engine, state, target=None, lam=1.0, w_div=1.0, layout='bchw', fused=True):
    """state, target: float32 fields (channels u, v, p), [B, 3, nx, ny] (layout='bchw') or channel-major [3, B, nx, ny]
    (layout='cm': the prediction's channels are contiguous fields as they come).  Returns (total, data, phys).

    fused (default): everything between the MLP and the residual kernels is the three passes of ops.PinnHeadFn
    (csrc/pinn_kernels.hip) -- one autograd node; fused=False: the same graph from tensor ops around
    ResidualEngine.differentiable (~60 small launches per step; kept as the cross-check of the fused head)."""
    if fused:
        from .. import ops
        if layout == 'cm':
            c, B, nx, ny = state.shape
            flat = lambda t: t.reshape(1, c, B * nx * ny)
            out, st, tg = model.mlp.train_forward(flat(state)), flat(state), (flat(target) if target is not None else None)
        else:
            B, c, nx, ny = state.shape
            out, st, tg = model.mlp.train_forward(state), state, target
        cont = lambda t: t if t is None or t.is_contiguous() else t.contiguous()
        return ops.PinnHeadFn.apply(cont(out), cont(st), cont(tg), engine.residual_spec(), (B, nx, ny), lam, w_div)
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


def train_step(model, engine, optimizer, state, target=None, lam=1.0, w_div=1.0, bucket=None, layout='bchw', fused=True):
    """One optimiser step.  Data-parallel runs pass `bucket` (nns.data_parallel.FlatGradAllReduce over the model's
    parameters): each rank works on its shard of the batch and the gradients are averaged with ONE all-reduce."""
    if bucket is not None:
        bucket.zero_()
    else:
        optimizer.zero_grad(set_to_none=True)
    total, data, phys = physics_informed_loss(model, engine, state, target, lam, w_div, layout=layout, fused=fused)
    total.backward()
    if bucket is not None:
        bucket.reduce_()
    optimizer.step()
    return total.detach(), data.detach(), phys.detach()
