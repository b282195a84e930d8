"""Sparse variational GP layer with decoupled (pathwise) sampling -- operator API of
experiments/model/core/svpy.py, computed by HIP kernels.

state_dict keys are the reference's: ``kern.unconstrained_lengthscales``, ``kern.unconstrained_variance``,
``inducing_loc.optvar`` (M,D_in), ``Um.optvar`` (M,D_out), ``Us_sqrt.optvar`` (D_out, M(M+1)/2).
"""
import numpy as np
import torch

from ..misc import transforms
from ..misc.param import Param
from .kernels import RBF, DivergenceFreeKernel
from .noise import NumpyNoise
from ... import ops

jitter = 1e-5


class SVGP_Layer(torch.nn.Module):
    def __init__(self, D_in, D_out, M, S, q_diag=False, dimwise=True, device='cpu', kernel='RBF'):
        super().__init__()
        if kernel == 'RBF':
            self.kern = RBF(D_in, D_out, dimwise)
            self.dimwise = dimwise
        elif kernel == 'DF':
            self.kern = DivergenceFreeKernel(D_in, D_out)
            self.dimwise = False  # as the reference (svpy.py:62-64)
        else:
            raise SystemExit('Invalid kernel selection')
        self.kernel_n = kernel
        self.q_diag = q_diag
        self.D_out, self.D_in, self.M, self.S = D_out, D_in, M, S
        self.device = device
        # initial values exactly as svpy.py:76-86 (global numpy RNG, same draw order)
        self.inducing_loc = Param(np.random.normal(size=(M, D_in)), name='Inducing locations')
        self.Um = Param(np.random.normal(size=(M, D_out)) * 1e-1, name='Inducing distribution (mean)')
        if q_diag:   # svpy.py:79-82: diagonal scale (M,D_out) under a softplus
            self.Us_sqrt = Param(np.ones(shape=(M, D_out)) * 1e-3, transform=transforms.SoftPlus(),
                                 name='Inducing distribution (scale)')
            self._diag_index = torch.tensor([n * (n + 1) // 2 + n for n in range(M)], dtype=torch.long)
        else:
            self.Us_sqrt = Param(np.stack([np.eye(M)] * D_out) * 1e-3,
                                 transform=transforms.LowerTriangular(M, D_out, device=self.device),
                                 name='Inducing distribution (scale)')
        self.noise_source = NumpyNoise()
        self._next_noise = []
        self.cache = None
        self._width_pad = False                      # resolved on first use (the library is loaded lazily)

    @property
    def width_pad(self):
        """ops.WidthPad when (D_in, D_out) is not a compiled width (RBF: evaluated zero-padded at the next compiled one)."""
        if self._width_pad is False:
            self._width_pad = ops.width_pad(self.kernel_n, self.D_in, self.D_out)
        return self._width_pad

    def us_packed(self):
        """The inducing scale in the packed lower-triangular layout (D_out, M(M+1)/2) the kernels consume.  q_diag=False:
        the parameter itself.  q_diag=True (svpy.py:95-96,153-167): softplus(raw) scattered onto the packed diagonal --
        the triangular mat-vec, its gradient and the KL then reduce to the diagonal formulas of the reference."""
        if not self.q_diag:
            return self.Us_sqrt.optvar
        raw = self.Us_sqrt.optvar                                   # (M, D_out)
        idx = self._diag_index.to(raw.device)
        packed = torch.zeros(self.D_out, self.M * (self.M + 1) // 2, dtype=raw.dtype, device=raw.device)
        return packed.index_copy(1, idx, (torch.nn.functional.softplus(raw) + 1e-12).t())

    # -- randomness ---------------------------------------------------------------------------
    def set_noise(self, *noises):
        """Queue explicit draws (dicts, see core/noise.py): each build_cache() consumes one instead of drawing."""
        self._next_noise.extend(noises)

    def _take_noise(self):
        dev = self.inducing_loc.optvar.device
        if self._next_noise:
            nz = self._next_noise.pop(0)
            return {k: v.to(dev) for k, v in nz.items()}
        return self.noise_source.draw(self.kernel_n, self.D_in, self.D_out, self.M, self.S, dev,
                                      **({} if self.dimwise or self.kernel_n != 'RBF' else {'dimwise': False}))

    def _take_noise_draws(self, L):
        """L draws stacked along a leading axis, in the order L successive build_cache() calls would have consumed them."""
        dev = self.inducing_loc.optvar.device
        plain = self.dimwise or self.kernel_n != 'RBF'
        if not self._next_noise and plain and hasattr(self.noise_source, 'draw_n'):
            return self.noise_source.draw_n(self.kernel_n, self.D_in, self.D_out, self.M, self.S, dev, L)   # one launch for all draws
        per = [self._expand_shared(self._take_noise()) for _ in range(L)]
        return {k: torch.stack([nz[k] for nz in per]) for k in per[0]}

    def predraw(self, draws=None):
        """Draw the noise of the next build_cache(draws=...) NOW (on the current stream) and keep it for that call -- used with a
        device noise source whose launch also carries other draws of the step (DeviceNoise.reserve)."""
        if self._next_noise:
            return                                   # explicit draws are queued: they win, nothing is drawn
        self._predrawn = (draws, self._expand_shared(self._take_noise()) if draws is None else self._take_noise_draws(draws))

    def _expand_shared(self, nz):
        if nz['rff_eps'].dim() == 2:   # dimwise=False draws one frequency / phase set (kernels.py:118-124,131-132): repeat it per output
            nz = dict(nz, rff_eps=nz['rff_eps'].unsqueeze(-1).expand(-1, -1, self.D_out).contiguous(),
                      rff_u=nz['rff_u'].unsqueeze(-1).expand(-1, -1, self.D_out).contiguous())
        return nz

    # -- reference API ------------------------------------------------------------------------
    def sample_inducing(self):
        """One draw u ~ q(u) = N(m, S) in whitened form (svpy.py:88-101); returns (M,D_out)."""
        self.build_cache()
        return self.cache.u

    def _cache_inputs(self, noise=None, draws=None):
        """Everything the cache-build kernels read, produced by torch on the CURRENT stream: the draw(s), and the derived tensors
        of the operator variants (q_diag: softplus scale scattered onto the packed diagonal; dimwise=False: shared
        hyper-parameters / frequencies repeated per output).  Returned tensors are kept referenced by the cache."""
        if noise is not None and draws is not None:  # the L draws handed over stacked: a leading draw axis on every tensor
            if any(v.shape[0] != draws for v in noise.values()):
                raise ValueError('noise for %d draws must carry a leading axis of that length' % draws)
            nz = {k: v.to(self.inducing_loc.optvar.device) for k, v in noise.items()}
        else:
            if noise is not None:
                self._next_noise.insert(0, noise)
            pre, self._predrawn = getattr(self, '_predrawn', None), None
            if pre is not None and pre[0] == draws and not self._next_noise:
                nz = pre[1]
            else:
                nz = self._expand_shared(self._take_noise()) if draws is None else self._take_noise_draws(draws)
        raw_ell, raw_var = self.kern.raw_dimwise()
        params = (raw_ell.detach(), raw_var.detach(), self.inducing_loc.optvar.detach(), self.Um.optvar.detach(), self.us_packed().detach())
        pad = self.width_pad
        if pad is not None:
            if draws is not None:
                raise NotImplementedError('batched draws at a zero-padded latent width: the flow falls back to one build per draw')
            nz, params = pad.noise(nz), tuple(t.contiguous() for t in pad.params(*params))
        return nz, params

    def _launch_cache_build(self, nz, params, want_Lu=False):
        self.cache = ops.cache_build(self.kernel_n, *params, nz['eps_u'], nz['rff_w'], nz['rff_eps'], nz['rff_u'], want_Lu=want_Lu)
        self.cache.noise = nz
        self.cache.inputs = params      # alive for as long as a (side-stream) kernel may read them
        self.kern._set_cache(self.cache, nz)
        return self.cache

    def build_cache(self, noise=None, want_Lu=False, draws=None):
        """Fix one function draw: Fourier features, inducing sample, nu (svpy.py:103-121).  ``draws`` = L fixes L of them in one
        build (the L calls of odegpvae.py:41-43): K_uu is factored once, the cache's tensors carry a leading draw axis, and the
        attributes on ``kern`` show the last draw -- the state the reference's loop leaves behind."""
        nz, params = self._cache_inputs(noise, draws)
        return self._launch_cache_build(nz, params, want_Lu)

    def batched_draws_supported(self):
        return self.width_pad is None

    def prebuild_cache(self, draws=None):
        """Overlap mode (ops.set_overlap): draw the noise now and build the cache on the side stream, so that the
        Cholesky chain runs next to the encoder; Flow.forward picks it up with take_prebuilt_cache()."""
        if not ops.overlap_enabled():
            return
        # every tensor the side-stream kernels read is produced on the current stream BEFORE the fork (the side stream then
        # waits for it) and stays referenced by the cache until the step's join
        nz, params = self._cache_inputs(draws=draws)
        side = ops.fork_side_stream()
        ops.start_marker()
        with ops.launch_on(side):
            self._prebuilt = cache = self._launch_cache_build(nz, params)
        # the flow waits for THIS point of the side stream only; the gradient-independent half of the cache backward (L^-1 from the
        # factor) then follows on the same stream, behind the build it depends on -- one fork of the step's graph instead of two
        # (every fork / join of the captured step costs tens of microseconds at replay)
        self._prebuilt_ready = torch.cuda.Event()
        self._prebuilt_ready.record(side)
        cache.prepared = None
        if ops.PREPARE_WITH_PREBUILD and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            with ops.launch_on(side):
                cache.prepared = ops.cache_bwd_prepare(cache)

    def take_prebuilt_cache(self):
        cache = getattr(self, '_prebuilt', None)
        if cache is None:
            return None
        self._prebuilt = None
        torch.cuda.current_stream().wait_event(self._prebuilt_ready)
        if self.width_pad is not None:
            # the unpadded attribute slices are torch-native gathers of the cache's tensors: they launch on the current stream,
            # so they are taken here, after the join, not next to the side-stream kernels that write those tensors
            self.kern._set_cache(cache, cache.noise)
        return cache

    def forward(self, x):
        """f(x) = f_prior(x) + K(x,Z) nu for the cached draw (svpy.py:123-142)."""
        if self.cache is None:
            raise RuntimeError('call build_cache() first')
        pad = self.width_pad
        if pad is not None:
            return ops.rhs(self.cache, pad.state(x).contiguous(), mode=0)[:, :self.D_out]
        return ops.rhs(self.cache, x, mode=0)

    def build_conditional(self, x, full_cov=False):
        """q(f(x)) = N(m(x), Sigma(x)) with m = A^T Um, Sigma = K(x,x) + A^T (Us Us^T - I) A, A = L^-1 K(Z,x)
        (svpy.py:176-210): returns (mean (N,D_out), var (N,D_out)) or, with ``full_cov``, var (N,N,D_out).  RBF kernel (the
        reference's einsums assume its (D,M,M) layout); computed without autograd."""
        if self.kernel_n != 'RBF':
            raise NotImplementedError('build_conditional: the reference formulates it for the RBF kernel (svpy.py:189-209)')
        k = self.kern
        raw_ell, raw_var = k.raw_dimwise() if hasattr(k, 'raw_dimwise') else (k.unconstrained_lengthscales, k.unconstrained_variance)
        with torch.no_grad():
            # q_diag: the reference turns the (M,D) scale into (D,M,1) columns, so Us Us^T is the rank-one s s^T (svpy.py:194-195)
            Us = self.Us_sqrt().T.contiguous() if self.q_diag else self.us_packed()
            return ops.conditional(raw_ell.detach(), raw_var.detach(), self.inducing_loc.optvar.detach(), self.Um.optvar.detach(),
                                   Us.detach(), x.detach().to(self.inducing_loc.optvar.device, torch.float32), full_cov,
                                   us_rank1=self.q_diag)

    def kl(self):
        """KL(q(u) || N(0,I)) in whitened form (svpy.py:144-175)."""
        return ops.svgp_kl(self.Um.optvar, self.us_packed(), self.M)
