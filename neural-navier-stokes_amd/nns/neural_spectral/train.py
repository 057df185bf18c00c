"""The training driver of the reference's ``spectral_ode.py`` / ``spectral_ode2.py`` (``__main__``, :140-224): same
command-line flags, same data handling (first 100 steps, float32, batch of 1), Adam(lr=1e-3), checkpoint every 10
iterations with the same dict keys, final ``extrapolation.npy``.  The forward/backward runs on the fused HIP
kernels (``PDEFunc.loss``); ``--unfused-loss`` uses the reference's literal ``torch.norm(model(...) - obs)``."""
import argparse
import os

import numpy as np
import torch
import torch.optim as optim  # noqa: F401  (kept: checkpoints of either optimiser class load into the other)

from ..optim import Adam


def main(which='spectral_ode'):
    from . import spectral_ode, spectral_ode2
    mod = spectral_ode if which == 'spectral_ode' else spectral_ode2
    parser = argparse.ArgumentParser()
    parser.add_argument('--npz-path', type=str, default='../data/data_semi_implicit.npz')
    parser.add_argument('--out-dir', type=str, default='./checkpoints/%s' % which,
                        help='where to save checkpoints [default: ./checkpoints/%s]' % which)
    parser.add_argument('--n-iters', type=int, default=1000, help='default: 1000')
    parser.add_argument('--n-coeffs', type=int, default=10, help='default: 10')
    parser.add_argument('--gpu-device', type=int, default=0, help='default: 0')
    parser.add_argument('--unfused-loss', action='store_true')
    parser.add_argument('--no-graph', action='store_true', help='run every iteration eagerly instead of replaying ONE captured HIP graph of loss + backward')
    args = parser.parse_args()
    args.out_dir = '{}_{}'.format(args.out_dir, args.n_coeffs)
    if not os.path.isdir(args.out_dir):
        os.makedirs(args.out_dir)
    device = (torch.device('cuda:' + str(args.gpu_device)) if torch.cuda.is_available() else 'cpu')

    def load(limit):
        data = np.load(args.npz_path)
        u, v, p = (torch.from_numpy(data[k][:limit] if limit else data[k]).float() for k in ('u', 'v', 'p'))
        obs = torch.stack([u, v, p]).permute(1, 0, 2, 3).to(device)
        nt, nx, ny = obs.size(0), obs.size(2), obs.size(3)
        obs = obs.unsqueeze(1).contiguous()                     # add a batch size of 1
        return obs, obs[0], (torch.arange(nt) + 1).to(device), nx, ny

    obs, obs0, t, nx, ny = load(100)
    model = mod.PDEFunc(args.n_coeffs, nx, ny).to(device)
    # the reference's optim.Adam(model.parameters(), lr=1e-3) (spectral_ode.py:171): same update and state_dict, one HIP launch per step
    optimizer = Adam(model.parameters(), lr=1e-3)
    loss_meter, penalty_meter = mod.AverageMeter(), mod.AverageMeter()
    losses, penalties = [], []
    loss_fn = (lambda: torch.norm(model(obs0, t) - obs, p=2)) if args.unfused_loss else (lambda: model.loss(obs0, t, obs))
    graphed = None
    if device != 'cpu' and not args.no_graph:
        # the iteration runs the same kernels on the same buffers every time (the observations are fixed, :175-177): capture it once (nns/graphs.py)
        from ..graphs import GraphedBackward
        try:
            graphed = GraphedBackward(model.parameters(), loss_fn)
        except Exception as e:                                   # noqa: BLE001 -- a model whose step does not capture trains eagerly
            print('train: HIP graph capture of the iteration failed (%r): running eagerly' % (e,))
            for q in model.parameters():
                q.grad = None
    for itr in range(1, args.n_iters + 1):
        if graphed is not None:
            loss = graphed()
        else:
            optimizer.zero_grad()
            loss = loss_fn()
        if hasattr(model, 'diversity_penalty'):
            with torch.no_grad():
                penalty = 1. / model.diversity_penalty()
                penalty_meter.update(penalty.item())
                penalties.append(penalty.item())
        if graphed is None:
            loss.backward()
        optimizer.step()
        loss_meter.update(loss.item())
        losses.append(loss.item())
        if itr % 10 == 0:
            torch.save({
                'model_state_dict': model.state_dict(),
                'optimizer_state_dict': optimizer.state_dict(),
                'config': args,
                'losses': np.array(losses),
                'penalties': np.array(penalties),
            }, os.path.join(args.out_dir, 'checkpoint.pth.tar'))
    with torch.no_grad():
        obs, obs0, t, nx, ny = load(None)
        obs_pred = model(obs0, t).squeeze(1).cpu().detach().numpy()          # nt x 3 x nx x ny
    np.save(os.path.join(args.out_dir, 'extrapolation.npy'), obs_pred)
    return losses
