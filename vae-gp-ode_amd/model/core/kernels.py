"""GP covariance kernels of the latent ODE (operator API of experiments/model/core/kernels.py).

Same class names, constructor signatures, parameter names/shapes (``unconstrained_lengthscales``
(D_out,D_in), ``unconstrained_variance`` (D_out,)) and cached attributes (``nu``, ``rff_weights``,
``rff_omega``, ``rff_phase``) as the reference; the arithmetic runs in hand-written HIP kernels
(csrc/gp_cache.hip, csrc/gp_forward.hip) behind the C ABI of include/gpode.h.  Only the dimwise RBF
(reference default, main.py:63) and the divergence-free kernel are on the hot path.
"""
import torch
from torch import nn

from .. import misc  # noqa: F401
from ..misc.constraint_utils import invsoftplus, softplus
from ... import ops

jitter = 1e-5


class RBF(nn.Module):
    """Squared-exponential kernel with per-output-dimension hyper-parameters (kernels.py:29-195)."""
    kernel_id = 'RBF'

    def __init__(self, D_in, D_out=None, dimwise=False):
        super().__init__()
        self.D_in = D_in
        self.D_out = D_in if D_out is None else D_out
        self.dimwise = dimwise
        # dimwise=False (kernels.py:45-46): one lengthscale vector and one variance shared by all outputs
        self.unconstrained_lengthscales = nn.Parameter(torch.ones(self.D_out, self.D_in) if dimwise else torch.ones(self.D_in))
        self.unconstrained_variance = nn.Parameter(torch.ones(self.D_out) if dimwise else torch.ones(1))
        with torch.no_grad():  # class defaults of the reference (kernels.py:52-54)
            self.unconstrained_lengthscales.fill_(invsoftplus(torch.tensor(0.2)).item())
            self.unconstrained_variance.fill_(invsoftplus(torch.tensor(0.1)).item())
        self._cache = None

    @property
    def lengthscales(self):
        return softplus(self.unconstrained_lengthscales)

    @property
    def variance(self):
        return softplus(self.unconstrained_variance)

    def raw_dimwise(self):
        """(raw lengthscales (D_out,D_in), raw variances (D_out,)) as the kernels consume them.  The shared-parameter
        kernel is evaluated as the per-output kernel with the parameters repeated along the output axis (formula by
        formula the same arithmetic; autograd sums the per-output gradients back onto the shared parameters)."""
        if self.dimwise or self.kernel_id != 'RBF':
            return self.unconstrained_lengthscales, self.unconstrained_variance
        return (self.unconstrained_lengthscales.unsqueeze(0).expand(self.D_out, -1).contiguous(),
                self.unconstrained_variance.expand(self.D_out).contiguous())

    # -- cached per-draw state (set by SVGP_Layer.build_cache) ---------------------------------
    def _set_cache(self, cache, noise):
        self._cache = cache
        if cache.stacked:
            # L draws built together: the attributes show the LAST one, as after the reference's loop (odegpvae.py:41-43).  The
            # views are taken lazily -- indexing launches nothing, but keep the torch-native work off a side-stream build
            last = lambda t: t[-1]
            self.rff_weights = last(noise['rff_w'])
            if self.dimwise or self.kernel_id != 'RBF':
                self.rff_omega, self.rff_phase, self.nu = last(cache.omega), last(cache.phase), last(cache.nu)
            else:
                self.rff_omega, self.rff_phase = last(cache.omega)[..., 0], last(cache.phase)[..., 0]
                self.nu = last(cache.nu).reshape(self.D_out, -1).t()
            return
        self.rff_weights = noise['rff_w']
        if cache.Do != self.D_out or cache.Di != self.D_in:      # evaluated zero-padded (ops.WidthPad): expose the unpadded slices
            if ops.launching_on_side():                          # SVGP_Layer.take_prebuilt_cache() calls again after the join
                self.rff_omega = self.rff_phase = self.nu = None
                return
            pad = ops.width_pad(self.kernel_id, self.D_in, self.D_out)
            idx = pad.index(cache.omega.device)
            self.rff_weights = noise['rff_w'][:, :self.D_out]
            self.rff_omega = cache.omega.index_select(0, idx)[..., :self.D_out]
            self.rff_phase, self.nu = cache.phase[..., :self.D_out], cache.nu[:self.D_out]
        elif self.dimwise or self.kernel_id != 'RBF':
            self.rff_omega, self.rff_phase, self.nu = cache.omega, cache.phase, cache.nu
        else:   # attribute layouts of the reference's non-dimwise branch (kernels.py:118-132,164-172)
            self.rff_omega, self.rff_phase = cache.omega[..., 0], cache.phase[..., 0]
            self.nu = cache.nu.reshape(self.D_out, -1).t()

    def _need_cache(self):
        if self._cache is None:
            raise RuntimeError('call SVGP_Layer.build_cache() first (flow.py:22-25 does it before every odeint)')
        return self._cache

    def rff_forward(self, x, S=None):
        """Prior sample f_prior(x) from the cached Fourier features (kernels.py:140-153 / :319-351)."""
        return self._rhs(x, 1)

    def f_update(self, x, x2=None):
        """Pathwise update K(x,Z) nu with the cached nu (kernels.py:174-181 / :390-393)."""
        return self._rhs(x, 2)

    def _rhs(self, x, mode):
        c = self._need_cache()
        if c.Di != self.D_in or c.Do != self.D_out:
            pad = ops.width_pad(self.kernel_id, self.D_in, self.D_out)
            return ops.rhs(c, pad.state(x).contiguous(), mode=mode)[:, :self.D_out]
        return ops.rhs(c, x, mode=mode)

    def forward(self, X, X2=None):
        return self.K(X, X2)

    def K(self, X, X2=None):
        """K(X, X2): (D_out,N,M) for RBF, (N*D, M*D) for DF (kernels.py:98-110 / :289-303)."""
        ell, var = self.raw_dimwise()
        Kd = ops.kernel_matrix(self.kernel_id, ell.detach(), var.detach(), X, X2)
        return Kd if (self.dimwise or self.kernel_id != 'RBF') else Kd[0]


class DivergenceFreeKernel(RBF):
    """Matrix-valued divergence-free kernel built on the dimwise RBF parameters (kernels.py:201-393)."""
    kernel_id = 'DF'

    def __init__(self, D_in, D_out):
        if D_in != D_out:
            raise ValueError('DivergenceFreeKernel needs D_in == D_out (kernels.py:259-262)')
        super().__init__(D_in=D_in, D_out=D_out, dimwise=True)
