"""Variable transforms that define the state_dict layout of constrained parameters
(mirrors experiments/model/misc/transforms.py:8-81).

LowerTriangular packs each (N,N) matrix as its N(N+1)/2 lower-triangular entries in row-major
``np.tril_indices`` order -- that packed layout IS the checkpoint format of ``Us_sqrt.optvar`` and is
consumed directly by the HIP kernels (no unpacking on the hot path)."""
import numpy as np
import torch
import torch.nn.functional as F

from .settings import settings


class Identity:
    def __str__(self):
        return 'Identity transformation'

    def forward(self, x):
        return x

    backward = forward_tensor = backward_tensor = forward


class SoftPlus:
    def __init__(self, lower=1e-12):
        self._lower = lower

    def __str__(self):
        return 'Softplus transformation'

    def forward(self, x):
        return np.logaddexp(0, x) + self._lower

    def forward_tensor(self, x):
        return F.softplus(x) + self._lower

    def backward(self, y):
        ys = np.maximum(y - self._lower, np.finfo(settings.numpy_float).eps)
        return ys + np.log(-np.expm1(-ys))

    def backward_tensor(self, y):
        ys = torch.clamp_min(y - self._lower, torch.finfo(y.dtype).eps)
        return ys + torch.log(-torch.expm1(-ys))


class LowerTriangular:
    def __init__(self, N, num_matrices=1, device='cpu'):
        self.N = N
        self.num_matrices = num_matrices
        self.device = device
        r, c = np.tril_indices(N, 0)
        self._rows, self._cols = r, c
        self._idx_cache = {}

    def __str__(self):
        return 'Lower cholesky transformation'

    def forward(self, x):
        out = np.zeros((self.num_matrices, self.N, self.N), dtype=settings.numpy_float)
        out[:, self._rows, self._cols] = x
        return out

    def backward(self, y):
        return np.stack([m[self._rows, self._cols] for m in y])

    def _index(self, device):
        key = str(device)
        if key not in self._idx_cache:
            self._idx_cache[key] = (torch.as_tensor(self._rows, device=device), torch.as_tensor(self._cols, device=device))
        return self._idx_cache[key]

    def forward_tensor(self, x):
        r, c = self._index(x.device)
        out = torch.zeros((self.num_matrices, self.N, self.N), dtype=x.dtype, device=x.device)
        out[:, r, c] = x
        return out

    def backward_tensor(self, y):
        r, c = self._index(y.device)
        return y[:, r, c]
