"""``Adam``: torch.optim.Adam's update (the optimiser of every training loop of the reference: src/neural_spectral/spectral_ode.py:171,
spectral_ode2.py:159, rnn.py:90, spectral_rnn.py:131) as ONE HIP launch over all parameter tensors (nns_adam_step_f32,
csrc/optim_kernels.hip) instead of the seven launches of torch's foreach implementation.

Same constructor arguments, same ``state_dict()`` layout (per parameter: ``step`` -- a float32 scalar tensor as in torch >= 1.12 --,
``exp_avg``, ``exp_avg_sq``), so an ``optimizer_state_dict`` in a ``checkpoint.pth.tar`` loads into either class.  float32 parameters on
the HIP device only (the product has no CPU path); ``amsgrad`` is not implemented."""
import ctypes as C

import torch

from . import _lib
from ._lib import check


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, maximize=False):
        if amsgrad:
            raise NotImplementedError("nns.optim.Adam: amsgrad is not implemented (use torch.optim.Adam)")
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("nns.optim.Adam: invalid hyper-parameters lr=%r betas=%r eps=%r weight_decay=%r" % (lr, betas, eps, weight_decay))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=maximize))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            ps = [p for p in group['params'] if p.grad is not None]
            if not ps:
                continue
            steps = set()
            for p in ps:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError("nns.optim.Adam: parameters must be contiguous float32 tensors on the HIP device (got %s on %s)" % (p.dtype, p.device))
                if p.grad.is_sparse or p.grad.dtype != torch.float32:
                    raise RuntimeError("nns.optim.Adam: dense float32 gradients only")
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.zeros((), dtype=torch.float32)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['step'] += 1
                steps.add(int(st['step'].item()))            # a host tensor (torch's default for a non-capturable step): no device read-back
            if len(steps) != 1:                              # parameters added to the group later: one launch per step count
                by = {}
                for p in ps:
                    by.setdefault(int(self.state[p]['step'].item()), []).append(p)
            else:
                by = {steps.pop(): ps}
            for step, plist in by.items():
                n = len(plist)
                tab = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
                grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in plist]
                sizes = (C.c_long * n)(*[p.numel() for p in plist])
                check(_lib.lib().nns_adam_step_f32(tab(plist), tab(grads), tab([self.state[p]['exp_avg'] for p in plist]),
                                                   tab([self.state[p]['exp_avg_sq'] for p in plist]), sizes, n, group['lr'], group['betas'][0],
                                                   group['betas'][1], group['eps'], group['weight_decay'], step, int(bool(group['maximize'])),
                                                   torch.cuda.current_stream().cuda_stream), 'nns_adam_step_f32')
        return loss
