"""Constrained parameter wrapper (mirrors experiments/model/misc/param.py:7-28): the trainable tensor is
``optvar`` (unconstrained); calling the module returns the constrained value."""
import torch

from . import transforms
from .settings import settings


class Param(torch.nn.Module):
    def __init__(self, value, transform=None, name='var'):
        super().__init__()
        self.transform = transform if transform is not None else transforms.Identity()
        self.name = name
        raw = self.transform.backward(value)
        self.optvar = torch.nn.Parameter(torch.tensor(data=raw, dtype=settings.torch_float, device=settings.device))

    def __call__(self):
        return self.transform.forward_tensor(self.optvar)

    def __repr__(self):
        return '{} parameter with {}'.format(self.name, self.transform)
