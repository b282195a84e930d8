"""Kernel hyper-parameter initialisation (API of experiments/model/core/initialization.py:5-22).  The kernel is reached
through the attribute path the reference's callers use (``model.flow.odefunc.diffeq.kern``); values are given in the
constrained space (lengthscale, variance) and stored through the inverse softplus."""
import torch

from ..misc.constraint_utils import invsoftplus


def _set_constant(param, value, trainable):
    with torch.no_grad():
        param.copy_(invsoftplus(torch.full_like(param, float(value))))
    param.requires_grad_(trainable)


def initialize_and_fix_kernel_parameters(model, lengthscale_value=1.25, variance_value=0.5, fix=False):
    """Every lengthscale := lengthscale_value, every variance := variance_value; ``fix=True`` freezes both."""
    kern = model.flow.odefunc.diffeq.kern
    _set_constant(kern.unconstrained_lengthscales, lengthscale_value, not fix)
    _set_constant(kern.unconstrained_variance, variance_value, not fix)
    return model
