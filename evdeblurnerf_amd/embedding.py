"""Mirror of the reference ``networks/embedding.py`` Embedder (:65-98) / get_embedder (:101-115)."""
from __future__ import annotations

import torch

from . import _lib as L


class Embedder(torch.nn.Module):
    def __init__(self, multires: int, input_dims: int = 3):
        super().__init__()
        self.multires = multires
        self.input_dims = input_dims
        self.out_dim = input_dims * (1 + 2 * multires)

    def forward(self, inputs):
        sh = inputs.shape
        x = inputs.reshape(-1, sh[-1]).contiguous().float()
        out = torch.empty((x.shape[0], self.out_dim), dtype=torch.float32, device=x.device)
        L.check(L.lib().evd_embed(L.ptr(x), x.shape[0], sh[-1], self.multires, L.ptr(out), L.stream_ptr()), "evd_embed")
        return out.reshape(*sh[:-1], self.out_dim)


def get_embedder(multires, i=0, input_dim=3):
    if i == -1:
        return torch.nn.Identity(), 3
    e = Embedder(multires, input_dim)
    return e, e.out_dim
