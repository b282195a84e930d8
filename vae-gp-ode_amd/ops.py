"""Tensor-level wrappers over the C ABI: torch owns device memory and the stream, HIP does the work.

Every function takes CUDA(=HIP) fp32 tensors, hands raw device pointers + the CURRENT torch stream to
libgpode_hip.so and returns torch tensors.  Nothing here computes on the CPU.
"""
import ctypes

import torch

from . import _lib

KERNEL_ID = {'RBF': 0, 'DF': 1}
METHOD_ID = {'euler': 0, 'rk4': 1}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _chk(t, name, shape=None):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise _lib.GpodeError('%s must be a CUDA/HIP tensor (the HIP path has no CPU fallback)' % name)
    if t.dtype != torch.float32:
        raise _lib.GpodeError('%s must be float32, got %s' % (name, t.dtype))
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise _lib.GpodeError('%s: expected shape %s, got %s' % (name, tuple(shape), tuple(t.shape)))
    return t.contiguous()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def cache_sizes(kernel, Di, Do, M, S):
    p, w = ctypes.c_size_t(0), ctypes.c_size_t(0)
    _lib.call('gpode_cache_sizes', KERNEL_ID[kernel], Di, Do, M, S, ctypes.byref(p), ctypes.byref(w))
    return p.value, w.value


class GPCache:
    """Per-draw cache: the lane-major ``pack`` the kernels consume, plus the attributes the reference
    caches on ``kern`` (kernels.py:134-137,172)."""
    __slots__ = ('kernel', 'Di', 'Do', 'M', 'S', 'pack', 'ws', 'ell', 'var', 'omega', 'phase', 'u', 'Lu', 'nu',
                 'u_prior')

    def check_factorisation(self):
        """Raise like torch.linalg.cholesky does when K_uu + jitter*I is not positive definite
        (kernels.py:163 / :384).  Synchronises the stream."""
        info = ctypes.c_int(0)
        _lib.call('gpode_cache_info', _ptr(self.ws), ctypes.byref(info), _stream())
        if info.value & 1:
            raise _lib.GpodeError('linalg.cholesky: K_uu + jitter*I is not positive-definite')


def cache_build(kernel, raw_ell, raw_var, Z, Um, Us_packed, eps_u, rff_w, rff_eps, rff_u, want_Lu=False):
    """SVGP_Layer.build_cache (svpy.py:103-121) on the GPU."""
    Do, Di = raw_ell.shape
    M = Z.shape[0]
    S = rff_eps.shape[1]
    raw_ell = _chk(raw_ell, 'raw_ell', (Do, Di)); raw_var = _chk(raw_var, 'raw_var', (Do,))
    Z = _chk(Z, 'Z', (M, Di)); Um = _chk(Um, 'Um', (M, Do))
    Us_packed = _chk(Us_packed, 'Us_packed', (Do, M * (M + 1) // 2))
    eps_u = _chk(eps_u, 'eps_u', (M, Do))
    rff_w = _chk(rff_w, 'rff_w', (S if kernel == 'RBF' else 2 * S, Do))
    rff_eps = _chk(rff_eps, 'rff_eps', (Di, S, Do)); rff_u = _chk(rff_u, 'rff_u', (1, S, Do))
    pf, wf = cache_sizes(kernel, Di, Do, M, S)
    dev = Z.device
    new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    c = GPCache()
    c.kernel, c.Di, c.Do, c.M, c.S = kernel, Di, Do, M, S
    c.pack, c.ws = new(pf), new(wf)
    c.ell, c.var, c.omega, c.phase, c.u = new(Do, Di), new(Do), new(Di, S, Do), new(1, S, Do), new(M, Do)
    c.u_prior = new(M, Do)
    if kernel == 'RBF':
        c.nu = new(Do, M, 1)
        c.Lu = new(Do, M, M) if want_Lu else None
    else:
        c.nu = new(M * Do, 1)
        c.Lu = new(M * Do, M * Do) if want_Lu else None
    _lib.call('gpode_cache_build_fwd', KERNEL_ID[kernel], Di, Do, M, S,
              _ptr(raw_ell), _ptr(raw_var), _ptr(Z), _ptr(Um), _ptr(Us_packed),
              _ptr(eps_u), _ptr(rff_w), _ptr(rff_eps), _ptr(rff_u),
              _ptr(c.pack), _ptr(c.ws), _ptr(c.ell), _ptr(c.var), _ptr(c.omega), _ptr(c.phase), _ptr(c.u),
              _ptr(c.Lu), _ptr(c.nu), _ptr(c.u_prior), _stream())
    return c


def rhs(cache, x, mode=0):
    """SVGP_Layer.forward (svpy.py:123-142): x (N,Di) -> f (N,Do). mode 1: prior only, 2: update only."""
    x = _chk(x, 'x')
    if x.dim() != 2 or x.shape[1] != cache.Di:
        raise _lib.GpodeError('x must be (N,%d), got %s' % (cache.Di, tuple(x.shape)))
    N = x.shape[0]
    f = torch.empty((N, cache.Do), dtype=torch.float32, device=x.device)
    _lib.call('gpode_rhs_fwd', KERNEL_ID[cache.kernel], cache.Di, cache.Do, cache.M, cache.S, _ptr(cache.pack),
              _ptr(x), N, _ptr(f), mode, _stream())
    return f


def rollout(cache, z0, ts, order, method):
    """Flow.forward (flow.py:68-86) for a built cache: z0 (N,D), ts (T,) -> zt (N,T,D)."""
    if method not in METHOD_ID:
        raise _lib.GpodeError("solver '%s' is not a fixed-grid method of this build (euler, rk4)" % method)
    z0 = _chk(z0, 'z0'); ts = _chk(ts, 'ts')
    N, D = z0.shape
    if D != cache.Di or D != order * cache.Do:
        raise _lib.GpodeError('state dim %d must equal D_in=%d = order*D_out=%d' % (D, cache.Di, order * cache.Do))
    T = ts.shape[0]
    zt = torch.empty((N, T, D), dtype=torch.float32, device=z0.device)
    _lib.call('gpode_rollout_fwd', KERNEL_ID[cache.kernel], order, METHOD_ID[method], cache.Di, cache.Do, cache.M,
              cache.S, _ptr(cache.pack), _ptr(z0), _ptr(ts), N, T, _ptr(zt), _stream())
    return zt
