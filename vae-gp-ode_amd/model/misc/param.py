"""Constrained parameter: the trainable tensor is the UNCONSTRAINED ``optvar`` (that is what the optimiser, the checkpoint
and the HIP kernels see); evaluating the module maps it through the transform.  API of experiments/model/misc/param.py:7-28
(``Param(value, transform, name)``, ``.optvar``, ``param()``), which the reference borrows from GPflow's Parameter."""
import numpy as np
import torch

from . import transforms
from .settings import settings


class Param(torch.nn.Module):
    def __init__(self, value, transform=None, name='var'):
        torch.nn.Module.__init__(self)
        self.name = name
        self.transform = transforms.Identity() if transform is None else transform
        unconstrained = np.asarray(self.transform.backward(value))       # constrained initial value -> optimisation variable
        self.register_parameter('optvar', torch.nn.Parameter(
            torch.as_tensor(unconstrained, dtype=settings.torch_float).to(settings.device)))

    def forward(self):
        """The constrained value (differentiable w.r.t. ``optvar``)."""
        return self.transform.forward_tensor(self.optvar)

    def __repr__(self):
        return '%s parameter with %s' % (self.name, self.transform)
