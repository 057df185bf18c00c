"""Black-box next-frame baseline with the reference's class surface (src/neural_spectral/rnn.py:13-40; SURVEY.md section 8 (f)
rank 4): a GRU over flattened (u, v, p) frames and a two-layer read-out.  A comparison baseline on torch modules (MIOpen GRU,
rocBLAS linears) -- nothing here is on the residual hot path.  What this file owns is the state-dict contract (`gru.*`,
`linear.0.*`, `linear.2.*`, so reference checkpoints load with strict=True) and batch-safe shapes: the read-out is applied
to the [mb, nt, hidden] sequence as it stands (the reference flattens with a `.view` that only works for mb = 1), and the
roll-out writes into one preallocated host tensor."""
import torch
from torch import nn


def _readout(hidden, frame):
    return nn.Sequential(nn.Linear(hidden, hidden), nn.ReLU(), nn.Linear(hidden, frame))


class RNN(nn.Module):
    def __init__(self, input_dim, hidden_dim=256):
        super().__init__()
        self.input_dim, self.hidden_dim = input_dim, hidden_dim
        self.gru = nn.GRU(input_dim, hidden_dim, batch_first=True)
        self.linear = _readout(hidden_dim, input_dim)

    def forward(self, obs_seq):
        """obs_seq [mb, nt, input_dim] -> (next-frame predictions [mb, nt, input_dim], final hidden state [1, mb, hidden])."""
        states, last = self.gru(obs_seq)
        return self.linear(states), last

    @torch.no_grad()
    def extrapolate(self, obs, T_extrapolate):
        """Autoregressive roll-out from the frame(s) obs [mb, 1, input_dim]: [mb, T_extrapolate, input_dim] on the host."""
        frames = torch.empty(obs.size(0), T_extrapolate, self.input_dim, dtype=obs.dtype)
        frame, hidden = obs, None
        for k in range(T_extrapolate):
            state, hidden = self.gru(frame, hidden)
            frame = self.linear(state)
            frames[:, k] = frame[:, 0].cpu()
        return frames
