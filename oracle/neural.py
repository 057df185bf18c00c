"""Oracle (test infrastructure only): the neural_spectral field predictor.

Plain torch-CPU restatement (explicit parameter tensors, autograd for gradients, any dtype --
float64 is the checker's default) of:

  * ODEFunc           src/neural_spectral/spectral_ode.py:14-34   Linear-ReLU-Linear-ELU-Linear
  * Euler/RK2/RK4     src/neural_spectral/anode/scheme.py:21-42, time_stepper.py:35-45
                      (dt = 1/Nt, returns y_1..y_Nt stacked, y_0 excluded)
  * PDEFunc.forward   src/neural_spectral/spectral_ode.py:62-81   (shared init coeffs, 3K ODE)
                      src/neural_spectral/spectral_ode2.py:69-107 (three K-dim ODEs)
  * loss              src/neural_spectral/spectral_ode.py:182      torch.norm(pred - obs, 2)
  * diversity penalty src/neural_spectral/spectral_ode.py:90-97
  * BasisFunc         src/neural_spectral/spectral_ode.py:100-119  per-pixel MLP (1x1 convs)

The ANODE "checkpointing adjoint" (anode/adjoint.py:38-70) recomputes the forward under grad and
differentiates it, so its gradients equal plain autograd through the integrator -- which is what
this oracle computes.
"""
import torch
import torch.nn.functional as Fn


def odefunc(mlp, y):
    """mlp = (W0 [H,K], b0 [H], W1 [H,H], b1 [H], W2 [K,H], b2 [K]); y [mb, K]."""
    W0, b0, W1, b1, W2, b2 = mlp
    h = torch.relu(y @ W0.t() + b0)
    h = Fn.elu(h @ W1.t() + b1)
    return h @ W2.t() + b2


def integrate(mlp, z0, Nt, method='RK4'):
    """Returns [Nt, mb, K].  time_stepper.py:35-45 / scheme.py:21-42."""
    dt = 1. / float(Nt)
    f = lambda y: odefunc(mlp, y)
    y = z0
    out = []
    for _ in range(Nt):
        if method == 'Euler':
            y = y + dt * f(y)
        elif method == 'RK2':
            k1 = dt * f(y)
            k2 = dt * f(y + 1.0 / 2.0 * k1)
            y = y + k2
        elif method == 'RK4':
            k1 = dt * f(y)
            k2 = dt * f(y + 1.0 / 2.0 * k1)
            k3 = dt * f(y + 1.0 / 2.0 * k2)
            k4 = dt * f(y + k3)
            y = y + 1.0 / 6.0 * k1 + 1.0 / 3.0 * k2 + 1.0 / 3.0 * k3 + 1.0 / 6.0 * k4
        else:
            raise ValueError(method)
        out.append(y)
    return torch.stack(out)


def pde_forward(init_coeffs, mlp, basis, mb, nt, method='RK4'):
    """spectral_ode.PDEFunc.forward (:62-81).  init_coeffs [3K] (k-major, channel-minor: the
    ``view(nt, mb, K, 3)`` at :71), basis [K, 3, nx, ny] -> [nt, mb, 3, nx, ny]."""
    K = basis.shape[0]
    coeff = integrate(mlp, init_coeffs.unsqueeze(0).repeat(mb, 1), nt, method)
    coeff = coeff.view(nt, mb, K, 3)
    return torch.einsum('tbkc,kcxy->tbcxy', coeff, basis), coeff


def pde2_forward(inits, mlps, bases, mb, nt, method='RK4'):
    """spectral_ode2.PDEFunc.forward (:69-107).  inits = 3 x [K]; mlps = 3 x mlp; bases =
    3 x [K, nx, ny]  ->  [nt, mb, 3, nx, ny]."""
    chans = []
    for init, mlp, basis in zip(inits, mlps, bases):
        coeff = integrate(mlp, init.unsqueeze(0).repeat(mb, 1), nt, method)     # [nt, mb, K]
        chans.append(torch.einsum('tbk,kxy->tbxy', coeff, basis))
    return torch.stack(chans, dim=2)


def loss_fn(pred, obs):
    """spectral_ode.py:182 -- Frobenius norm over all elements."""
    return torch.sqrt(torch.sum((pred - obs) ** 2))


def diversity_penalty(basis):
    """spectral_ode.py:83-97: 1 / sum_{i<=j} ||W_i - W_j||_2 over flattened bases."""
    K = basis.shape[0]
    W = basis.reshape(K, -1)
    tot = 0
    for i in range(K):
        for j in range(i, K):
            tot = tot + torch.norm(W[i] - W[j], p=2)
    return 1. / tot


def _bf16_round(t):
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


def pixel_mlp_backward(weights, biases, grid, gy, bf16=False):
    """Reverse-mode of pixel_mlp by hand (what autograd does for the Conv2d/ReLU stack of :100-119):
    returns (gx, [gW_l], [gb_l]).  Used to check the fused HIP backward; itself checked against autograd.
    bf16=True emulates the kernel's operand rounding (weights, layer inputs and deltas to bfloat16 before every
    product, exact accumulation, ReLU mask from the rounded layer input) so that the comparison can be tight."""
    rnd = _bf16_round if bf16 else (lambda t: t)
    L = len(weights)
    Wr = [rnd(W) for W in weights]
    ins = []
    h = grid
    for l in range(L):
        ins.append(rnd(h))
        h = torch.einsum('oc,bcxy->boxy', Wr[l], ins[l]) + biases[l][None, :, None, None]
        if l < L - 1:
            h = torch.relu(h)
    d = gy
    gWs, gbs = [None] * L, [None] * L
    for l in range(L - 1, -1, -1):
        dr = rnd(d)
        gWs[l] = torch.einsum('boxy,bcxy->oc', dr, ins[l])
        gbs[l] = dr.sum(dim=(0, 2, 3))
        d = torch.einsum('oc,boxy->bcxy', Wr[l], dr)
        if l > 0:
            d = d * (ins[l] > 0).to(d.dtype)
    return d, gWs, gbs


def pixel_mlp(weights, biases, grid):
    """BasisFunc (:100-119) generalised to any depth: 1x1 convs == per-pixel linears with ReLU
    between layers, none after the last.  weights[l] [C_out, C_in]; grid [mb, C_in0, nx, ny]."""
    h = grid
    L = len(weights)
    for l, (W, b) in enumerate(zip(weights, biases)):
        h = torch.einsum('oc,bcxy->boxy', W, h) + b[None, :, None, None]
        if l < L - 1:
            h = torch.relu(h)
    return h
