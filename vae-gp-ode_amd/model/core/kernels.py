"""GP covariance kernels of the latent ODE (operator API of experiments/model/core/kernels.py).

Same class names, constructor signatures, parameter names/shapes (``unconstrained_lengthscales``
(D_out,D_in), ``unconstrained_variance`` (D_out,)) and cached attributes (``nu``, ``rff_weights``,
``rff_omega``, ``rff_phase``) as the reference; the arithmetic runs in hand-written HIP kernels
(csrc/gp_cache.hip, csrc/gp_forward.hip) behind the C ABI of include/gpode.h.  Only the dimwise RBF
(reference default, main.py:63) and the divergence-free kernel are on the hot path.
"""
import numpy as np
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
        self._kern_nu = None
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

    # -- the kernel's own per-draw methods (a caller that keeps its own SVGP_Layer binds these) -------------------------------
    def _draw(self, shape, seed=None, uniform=False):
        """The reference's sample_normal / sample_uniform (kernels.py:13-26): a fresh RandomState per call for the normals (so
        --seed does not reach them), the global numpy RNG for an unseeded uniform."""
        if uniform and seed is None:
            return torch.tensor(np.random.uniform(low=0.0, high=1.0, size=shape).astype(np.float32))
        rng = np.random.RandomState() if seed is None else np.random.RandomState(seed)
        return torch.tensor((rng.uniform(low=0.0, high=1.0, size=shape) if uniform else rng.normal(size=shape)).astype(np.float32))

    def _per_output(self, eps, u):
        """Shared draws of the non-dimwise kernel (D_in,S) / (1,S) repeated per output, as the kernels consume them."""
        if eps.dim() == 2:
            eps = eps.unsqueeze(-1).expand(-1, -1, self.D_out).contiguous()
        if u is not None and u.dim() == 2:
            u = u.unsqueeze(-1).expand(-1, -1, self.D_out).contiguous()
        return eps, u

    def _kern_cache(self, rff_w, rff_eps, rff_u, device):
        ell, var = self.raw_dimwise()
        eps3, u3 = self._per_output(rff_eps.to(device), rff_u.to(device))
        return ops.kern_cache(self.kernel_id, ell.detach(), var.detach(), rff_w.to(device), eps3, u3)

    def sample_freq(self, S, seed=None, device='cpu'):
        """omega = eps / lengthscales^T, eps ~ N(0, 1): a sample from the spectral density (kernels.py:112-124).
        -> (D_in, S, D_out), or (D_in, S) for the shared-parameter kernel."""
        dim3 = self.dimwise or self.kernel_id != 'RBF'
        eps = self._draw((self.D_in, S, self.D_out) if dim3 else (self.D_in, S), seed)
        dev = self.unconstrained_lengthscales.device
        c = self._kern_cache(torch.zeros((S if self.kernel_id == 'RBF' else 2 * S), self.D_out), eps, torch.zeros(1, S, self.D_out), dev)
        return c.omega if dim3 else c.omega[..., 0]

    def build_cache(self, S, device=None, noise=None):
        """Fix the Fourier features of one prior draw: rff_weights, rff_omega, rff_phase (kernels.py:126-137 / :305-316);
        kern.rff_forward(x, S) then evaluates THIS draw.  ``noise``: dict(rff_w, rff_eps, rff_u) instead of drawing (the draw
        order without it is the reference's: weights, frequencies, phases)."""
        dim3 = self.dimwise or self.kernel_id != 'RBF'
        if noise is None:
            noise = dict(rff_w=self._draw((S if self.kernel_id == 'RBF' else 2 * S, self.D_out)),
                         rff_eps=self._draw((self.D_in, S, self.D_out) if dim3 else (self.D_in, S)),
                         rff_u=self._draw((1, S, self.D_out) if dim3 else (1, S), uniform=True))
        dev = self.unconstrained_lengthscales.device
        c = self._kern_cache(noise['rff_w'], noise['rff_eps'], noise['rff_u'], dev)
        self._cache, self._kern_nu = c, None
        self.rff_weights = c.noise['rff_w']
        self.rff_omega, self.rff_phase = (c.omega, c.phase) if dim3 else (c.omega[..., 0], c.phase[..., 0])
        self.nu = None
        return c

    def compute_nu(self, Ku, u_prior, inducing_val):
        """nu = K(Z,Z)^-1 (u - f_prior(Z)) in whitened form, from the caller's Ku / u_prior / u (kernels.py:155-172 / :376-387);
        sets ``self.nu`` ((D_out,M,1) dimwise RBF, (M,D_out) shared RBF, (M D,1) DF) for kern.f_update(x, x2)."""
        dev = self.unconstrained_lengthscales.device
        Kd = Ku.to(dev, torch.float32)
        if self.kernel_id == 'RBF' and not self.dimwise:      # one (M,M) matrix shared by all outputs
            Kd = Kd.unsqueeze(0).expand(self.D_out, -1, -1).contiguous()
        nu, ws = ops.compute_nu(self.kernel_id, self.D_in, self.D_out, Kd, u_prior.to(dev, torch.float32), inducing_val.to(dev, torch.float32))
        self._kern_nu, self._kern_nu_ws = nu, ws
        self.nu = nu if (self.dimwise or self.kernel_id != 'RBF') else nu.reshape(self.D_out, -1).t()
        return self.nu

    def _need_cache(self):
        if self._cache is None:
            raise RuntimeError('call SVGP_Layer.build_cache() first (flow.py:22-25 does it before every odeint)')
        return self._cache

    def rff_forward(self, x, S=None):
        """Prior sample f_prior(x) from the cached Fourier features (kernels.py:140-153 / :319-351)."""
        return self._rhs(x, 1)

    def f_update(self, x, x2=None):
        """Pathwise update K(x,Z) nu with the cached nu (kernels.py:174-181 / :390-393).  After kern.compute_nu(...) it is
        evaluated on the given x2 with that nu; inside a layer-built draw the cached (Z, nu) are used."""
        if getattr(self, '_kern_nu', None) is not None and x2 is not None:
            ell, var = self.raw_dimwise()
            return ops.f_update(self.kernel_id, ell.detach(), var.detach(), x, x2.to(x.device, torch.float32), self._kern_nu)
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
