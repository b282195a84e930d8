"""Positive-constraint helpers (mirrors experiments/model/misc/constraint_utils.py:5-13).

Host-side parameter utilities only (initialisation, logging of lengthscales): the hot path applies
softplus(+1e-12) inside the HIP cache-build kernel (csrc/gp_cache.hip:k_hyper)."""
import torch
import torch.nn.functional as F

LOWER = 1e-12


def softplus(x):
    return F.softplus(x) + LOWER


def invsoftplus(x):
    xs = torch.clamp_min(x - LOWER, torch.finfo(x.dtype).eps)
    return xs + torch.log(-torch.expm1(-xs))
