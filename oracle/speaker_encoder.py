"""CPU oracle of the speaker encoder network (TEST INFRASTRUCTURE -- only tests/ may import this).

Restates `SpeakerEncoder.forward` (`fam/quantiser/audio/speaker_encoder/model.py:54-57`) and the partial-utterance averaging of
`embed_utterance` (`:86-107`) in plain torch fp32 with explicit gate arithmetic (torch.nn.LSTM order i, f, g, o), so that it
does not share code with either the reference's nn.LSTM call or the HIP kernel.  Pinned by tests/golden/speaker_encoder.npz,
minted by running the reference class itself (tests/golden/make_golden.py).
"""
from typing import Dict

import torch


def lstm_forward(sd: Dict[str, torch.Tensor], mels: torch.Tensor, n_layers: int = 3) -> torch.Tensor:
    """(B, T, 40) -> h_T of the last layer, (B, H)."""
    x = mels.float()
    B, T, _ = x.shape
    h = None
    for l in range(n_layers):
        w_ih, w_hh = sd[f"lstm.weight_ih_l{l}"].float(), sd[f"lstm.weight_hh_l{l}"].float()
        bias = sd[f"lstm.bias_ih_l{l}"].float() + sd[f"lstm.bias_hh_l{l}"].float()
        H = w_hh.shape[1]
        h = torch.zeros(B, H)
        c = torch.zeros(B, H)
        ys = []
        gin = x @ w_ih.T + bias
        for t in range(T):
            g = gin[:, t] + h @ w_hh.T
            i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            ys.append(h)
        x = torch.stack(ys, dim=1)
    return h


def forward(sd: Dict[str, torch.Tensor], mels: torch.Tensor) -> torch.Tensor:
    """`model.py:54-57`: relu(linear(h_T)) / ||.||_2 per partial."""
    h = lstm_forward(sd, mels)
    e = torch.relu(h @ sd["linear.weight"].float().T + sd["linear.bias"].float())
    return e / torch.norm(e, dim=1, keepdim=True)
