"""Black-box next-frame baseline (reference: src/neural_spectral/rnn.py:13-40): a GRU over
flattened (u, v, p) frames followed by a two-layer MLP.  A comparison baseline, kept on torch modules (MIOpen GRU,
rocBLAS linears) -- nothing here is on the residual hot path.  Same class surface and state-dict names as the reference;
`forward` reshapes instead of `.view`, so batches larger than 1 work too (the reference's `.view` fails on them)."""
import torch
import torch.nn as nn


class RNN(nn.Module):
    def __init__(self, input_dim, hidden_dim=256):
        super().__init__()
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.gru = nn.GRU(self.input_dim, self.hidden_dim, batch_first=True)
        self.linear = nn.Sequential(
            nn.Linear(self.hidden_dim, self.hidden_dim),
            nn.ReLU(),
            nn.Linear(self.hidden_dim, self.input_dim))

    def forward(self, obs_seq):
        mb, nt = obs_seq.size(0), obs_seq.size(1)
        out_seq, gru_hid = self.gru(obs_seq, None)
        out_seq = self.linear(out_seq.reshape(mb * nt, -1))
        return out_seq.view(mb, nt, -1), gru_hid

    def extrapolate(self, obs, T_extrapolate):
        h0 = None
        out_extrapolate = []
        for _ in range(T_extrapolate):
            out, h0 = self.gru(obs, h0)
            obs = self.linear(out.squeeze(1)).unsqueeze(1)
            out_extrapolate.append(obs.cpu().detach())
        return torch.cat(out_extrapolate, dim=1)
