"""Developer probe: BASELINE config 2's training iteration with loss + backward captured in a HIP graph (torch.cuda.CUDAGraph) and replayed,
against the eager iteration.  The optimiser step stays outside the graph (its bias corrections are host constants of the step count)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'neural-navier-stokes_amd')):
    sys.path.insert(0, p)
import torch
from nns.neural_spectral.spectral_ode import PDEFunc
import nns.optim as nns_optim
K, n, nt = 10, 128, 100
torch.manual_seed(0)
m = PDEFunc(K, n, n).cuda()
obs = torch.randn(nt, 1, 3, n, n, device='cuda')
t = torch.arange(nt, device='cuda') + 1
opt = nns_optim.Adam(m.parameters(), lr=1e-3)
params = list(m.parameters())

def eager():
    opt.zero_grad()
    loss = m.loss(obs[0], t, obs)
    loss.backward()
    opt.step()
    return loss

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3

print('eager iteration: %.3f ms' % timeit(eager))
# capture: grads must exist as fixed buffers the graph writes into
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        m.loss(obs[0], t, obs).backward()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    static_loss = m.loss(obs[0], t, obs)
    static_loss.backward()
grads = [p.grad for p in params]

def graphed():
    g.replay()
    opt.step()
    return static_loss

print('graphed iteration: %.3f ms' % timeit(graphed))
# same numbers?
torch.manual_seed(0)
m2 = PDEFunc(K, n, n).cuda(); m2.load_state_dict(m.state_dict())
l_e = m2.loss(obs[0], t, obs); l_e.backward()
g.replay(); torch.cuda.synchronize()
print('loss eager %.6f graph %.6f; max grad diff %.2e' % (float(l_e), float(static_loss), max(float((a.grad - b.grad).abs().max()) for a, b in zip(m2.parameters(), params))))
