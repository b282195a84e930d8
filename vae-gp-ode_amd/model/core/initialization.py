"""Kernel hyper-parameter initialisation (mirrors experiments/model/core/initialization.py:5-22); reaches the
kernel through the same attribute path the reference's callers use."""
import torch

from ..misc.constraint_utils import invsoftplus


def initialize_and_fix_kernel_parameters(model, lengthscale_value=1.25, variance_value=0.5, fix=False):
    kern = model.flow.odefunc.diffeq.kern
    with torch.no_grad():
        kern.unconstrained_lengthscales.data = invsoftplus(
            lengthscale_value * torch.ones_like(kern.unconstrained_lengthscales.data))
        kern.unconstrained_variance.data = invsoftplus(
            variance_value * torch.ones_like(kern.unconstrained_variance.data))
    if fix:
        kern.unconstrained_lengthscales.requires_grad_(False)
        kern.unconstrained_variance.requires_grad_(False)
    return model
