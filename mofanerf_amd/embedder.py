"""``get_embedder`` with the reference's signature (models/model.py:48-63).

The renderer fuses the encoding into the first-layer kernel, so the returned callable is only used when a
caller wants the encoding itself; it runs the same HIP feature generator (``mofa_positional_encode``).
"""
import torch

from . import lib


class Embedder:
    def __init__(self, n_freqs: int):
        self.n_freqs, self.out_dim = n_freqs, 3 + 6 * n_freqs

    def embed(self, x: torch.Tensor) -> torch.Tensor:
        flat = x.detach().reshape(-1, 3).float().contiguous()
        out = torch.empty(flat.shape[0], self.out_dim, dtype=torch.float32, device=flat.device)
        lib.check(lib.load().mofa_positional_encode(lib.ptr(flat), flat.shape[0], self.n_freqs, lib.ptr(out),
                                                    lib.stream()), "mofa_positional_encode")
        return out.reshape(*x.shape[:-1], self.out_dim)

    __call__ = embed


def get_embedder(multires, i=0):
    if i == -1:
        return torch.nn.Identity(), 3
    e = Embedder(int(multires))
    return e, e.out_dim
