"""Host-side contracts of the round-4 training utilities (CPU only: no kernels run).  The product has no CPU path: nns.optim.Adam and
nns.graphs.GraphedBackward must refuse host tensors loudly instead of falling back to torch's optimiser / an eager loop."""
import sys

import pytest
import torch

from conftest import PKG

sys.path.insert(0, PKG)


def test_adam_refuses_host_parameters_and_unsupported_options():
    import nns.optim as nns_optim
    p = torch.nn.Parameter(torch.zeros(4))
    with pytest.raises(NotImplementedError, match='amsgrad'):
        nns_optim.Adam([p], amsgrad=True)
    with pytest.raises(ValueError, match='invalid hyper-parameters'):
        nns_optim.Adam([p], betas=(1.0, 0.999))
    opt = nns_optim.Adam([p], lr=1e-3)
    opt.step()                                            # no gradients yet: nothing to do, no library call
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match='HIP device'):
        opt.step()
    # the state layout is torch's: an optimiser state saved by torch.optim.Adam loads (and the other way round)
    q = torch.nn.Parameter(torch.zeros(4))
    t = torch.optim.Adam([q], lr=1e-3)
    q.grad = torch.ones(4)
    t.step()
    opt2 = nns_optim.Adam([torch.nn.Parameter(torch.zeros(4))], lr=5e-4)
    opt2.load_state_dict(t.state_dict())
    st = next(iter(opt2.state.values()))
    assert float(st['step']) == 1.0 and st['exp_avg'].shape == (4,) and opt2.param_groups[0]['lr'] == 1e-3
    t2 = torch.optim.Adam([torch.nn.Parameter(torch.zeros(4))])
    t2.load_state_dict(opt2.state_dict())
    assert float(next(iter(t2.state.values()))['step']) == 1.0


def test_graphed_backward_refuses_host_parameters():
    from nns.graphs import GraphedBackward
    p = torch.nn.Parameter(torch.zeros(3))
    with pytest.raises(RuntimeError, match='HIP device'):
        GraphedBackward([p], lambda: (p * p).sum())
    with pytest.raises(RuntimeError, match='HIP device'):
        GraphedBackward([], lambda: torch.zeros(()))
