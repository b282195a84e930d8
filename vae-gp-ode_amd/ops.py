"""Tensor-level wrappers over the C ABI: torch owns device memory and the stream, HIP does the work.

Every function takes CUDA(=HIP) fp32 tensors, hands raw device pointers + the CURRENT torch stream to
libgpode_hip.so and returns torch tensors.  Nothing here computes on the CPU.
"""
import ctypes
import os

import torch

from . import _lib

KERNEL_ID = {'RBF': 0, 'DF': 1}
METHOD_ID = {'euler': 0, 'rk4': 1, 'midpoint': 2}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _chk(t, name, shape=None):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise _lib.GpodeError('%s must be a CUDA/HIP tensor (the HIP path has no CPU fallback)' % name)
    if t.dtype != torch.float32:
        raise _lib.GpodeError('%s must be float32, got %s' % (name, t.dtype))
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise _lib.GpodeError('%s: expected shape %s, got %s' % (name, tuple(shape), tuple(t.shape)))
    if not t.is_contiguous() and _launch_override:
        # a torch-native copy launches on torch's current stream; inside launch_on() the kernels that read the copy do not
        with torch.cuda.stream(_launch_override[-1]):
            return t.contiguous()
    return t.contiguous()


# ---------------------------------------------------------------------------------------------
# Launch stream.  By default every kernel goes to torch's current stream.  In overlap mode (set_overlap(True)) the
# two serial, few-workgroup chains of the GP -- the cache build (Cholesky of K_uu, forward) and the cache backward
# (triangular inverse, chain rule to the raw parameters) -- are launched on a side stream, where they run next to
# the encoder's forward / backward kernels instead of in front of them.  Memory is always allocated on the current
# stream; buffers a side-stream kernel touches are kept referenced until join_side_stream().  Only THIS package's launches
# follow launch_on(): a torch-native op inside such a region (a fill, a gather, .contiguous()) still runs on torch's current
# stream, unordered against the side stream -- none may touch a buffer the side-stream kernels read or write
# (profiles/r02b_notes.txt: a torch.zeros there made the GP parameter gradients wobble from run to run).
# ---------------------------------------------------------------------------------------------
_launch_override = []          # stack of torch streams that _stream() returns instead of the current stream
_overlap = {'on': False, 'side': None, 'pending': [], 'forked': False, 'kl': None}


def _stream():
    if _launch_override:
        return ctypes.c_void_p(_launch_override[-1].cuda_stream)
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class launch_on:
    """Context: kernels of this package launch on ``stream`` (torch's current stream, and so allocation, unchanged)."""
    def __init__(self, stream):
        self.stream = stream

    def __enter__(self):
        _launch_override.append(self.stream)

    def __exit__(self, *exc):
        _launch_override.pop()


def launching_on_side():
    """True inside ``launch_on``: torch-native ops would run on another stream than this package's kernels."""
    return bool(_launch_override)


def set_overlap(on):
    """Run the GP cache build / cache backward on a side stream (see above).  Gradients of the GP parameters are then
    completed by join_side_stream(), which the optimiser and the gradient all-reduce of this package call themselves."""
    _overlap['on'] = bool(on)


def overlap_enabled():
    return _overlap['on']


def side_stream():
    if _overlap['side'] is None:
        _overlap['side'] = torch.cuda.Stream()
    return _overlap['side']


def fork_side_stream():
    """The side stream waits for everything queued so far on the current stream; returns it."""
    side = side_stream()
    side.wait_stream(torch.cuda.current_stream())
    _overlap['forked'] = True
    return side


def _main_marker():
    """One trivial kernel on the CURRENT stream (a 1-element fill of a private buffer).  Placed directly behind the fork of the
    backward pass: a captured training step whose main branch has no node of its own between that fork and the encoder's
    backward replays 0.07-0.15 ms slower (configs[1] 3.19 -> 3.12 ms, configs[0] 1.07 -> 0.93 ms; the first three nodes of
    the following replay then start ~55 us apart -- profiles/r02b_notes.txt).  Measured, not understood: a property of how the
    graph executor lays branches onto hardware queues."""
    if _NO_MARKER:
        return
    d = _overlap.get('marker')
    if d is None or d.device != torch.device('cuda', torch.cuda.current_device()):
        d = _overlap['marker'] = torch.zeros(1, dtype=torch.float32, device='cuda')
    d.zero_()


_START_MARKER = os.environ.get('GPODE_START_MARKER', '0') == '1'


def start_marker():
    """EXPERIMENT (GPODE_START_MARKER=1): the main branch's node created first at the fork that opens the step as well."""
    if _START_MARKER and _overlap['on']:
        _main_marker()


def defer_kl_grads(params, grads):
    """Overlap mode: the gradients of KL(q(u)||p(u)) w.r.t. (Um, Us) -- written at the very start of the backward pass -- are
    not handed to autograd (which would add the flow's gradients to them in a launch of its own per tensor) but kept for the
    flow's deferred backward, whose last kernel adds its own contribution in place.  True if taken."""
    if not _overlap['on'] or _overlap['kl'] is not None:
        return False
    _overlap['kl'] = (tuple(params), tuple(grads))
    return True


_NO_MARKER = os.environ.get('GPODE_NO_MARKER', '0') == '1'
PREPARE_WITH_PREBUILD = os.environ.get('GPODE_PREPARE_FORK', '0') != '1'
_HEARTBEAT = os.environ.get('GPODE_SIDE_HEARTBEAT', '0') == '1'


def side_heartbeat():
    """EXPERIMENT (GPODE_SIDE_HEARTBEAT=1): a trivial kernel on the side stream that waits for the current point of the main
    stream.  The side branch of the backward pass starts 60-115 us after the kernel it depends on has finished, and the longer
    the side queue has sat blocked the longer that takes; heartbeats keep its waits short."""
    if not (_HEARTBEAT and _overlap['on']):
        return
    side = side_stream()
    side.wait_stream(torch.cuda.current_stream())
    _main_marker()                                   # the main branch's node is created first: it keeps the parent's queue
    d = _overlap.get('hb')
    if d is None:
        d = _overlap['hb'] = torch.zeros(1, dtype=torch.float32, device='cuda')
    with torch.cuda.stream(side):
        d.zero_()
    _overlap['forked'] = True


def join_side_stream():
    """Current stream waits for the side stream; deferred parameter gradients are accumulated."""
    if not _overlap['forked'] and not _overlap['pending'] and _overlap['kl'] is None:
        return
    torch.cuda.current_stream().wait_stream(side_stream())
    _overlap['forked'] = False
    pend, _overlap['pending'] = _overlap['pending'], []
    kl, _overlap['kl'] = _overlap['kl'], None
    if kl is not None:                               # no flow backward took the KL gradients along: they are the parameters' own
        pend.append((kl[0], kl[1], None))
    with torch.no_grad():
        for params, grads, _keep in pend:
            for p, g in zip(params, grads):
                if p.grad is None:
                    p.grad = g
                else:
                    p.grad.add_(g)


def cache_sizes(kernel, Di, Do, M, S, ndraws=1):
    """(floats of ONE draw's pack, floats of the build workspace for ``ndraws`` draws that share the factor)."""
    p, w = ctypes.c_size_t(0), ctypes.c_size_t(0)
    _lib.call('gpode_cache_sizes_n', KERNEL_ID[kernel], Di, Do, M, S, ndraws, ctypes.byref(p), ctypes.byref(w))
    return p.value, w.value


class GPCache:
    """Per-draw cache: the lane-major ``pack`` the kernels consume, plus the attributes the reference
    caches on ``kern`` (kernels.py:134-137,172)."""
    __slots__ = ('kernel', 'Di', 'Do', 'M', 'S', 'pack', 'ws', 'ell', 'var', 'omega', 'phase', 'u', 'Lu', 'nu',
                 'u_prior', 'noise', 'inputs', 'nd', 'stacked', 'prepared')

    @property
    def lead(self):
        """() for a single draw; (L,) when L Monte-Carlo draws share this build (every per-draw tensor then has a leading draw axis)."""
        return (self.nd,) if self.stacked else ()

    def check_factorisation(self):
        """Raise like torch.linalg.cholesky does when K_uu + jitter*I is not positive definite
        (kernels.py:163 / :384).  Synchronises the stream."""
        info = ctypes.c_int(0)
        _lib.call('gpode_cache_info', _ptr(self.ws), ctypes.byref(info), _stream())
        if info.value & 1:
            raise _lib.GpodeError('linalg.cholesky: K_uu + jitter*I is not positive-definite')


    def pivot_range(self):
        """(min, max) diagonal entry of the draw's Cholesky factor(s): (max / min)^2 bounds cond(K_uu + jitter I) from below.
        Synchronises the stream."""
        mm = (ctypes.c_float * 2)()
        _lib.call('gpode_cache_pivots', _ptr(self.ws), mm, _stream())
        return float(mm[0]), float(mm[1])


BACKWARD_SOLVES = {'auto': 0, 'always': 1, 'never': 2}


def set_backward_solves(mode):
    """How the cache backward goes through the factor: 'auto' (triangular solves up to 192 rows, explicit inverse beyond -- the
    fastest on a well-conditioned K_uu), 'always' (solves up to 1216 rows: torch-grade accuracy on a rank-deficient K_uu), 'never'."""
    _lib.call('gpode_set_backward_solves', BACKWARD_SOLVES[mode])


def cache_build(kernel, raw_ell, raw_var, Z, Um, Us_packed, eps_u, rff_w, rff_eps, rff_u, want_Lu=False):
    """SVGP_Layer.build_cache (svpy.py:103-121) on the GPU.  Noise with a LEADING draw axis (eps_u (L,M,Do), rff_w (L,S,Do) ...)
    builds L Monte-Carlo draws at once: K_uu + jitter I is factored ONCE (it depends on the parameters only, kernels.py:163 / :384)
    and the L right-hand sides are solved as a block; pack, omega, phase, u, nu, u_prior then carry the draw axis too."""
    Do, Di = raw_ell.shape
    M = Z.shape[0]
    stacked = eps_u.dim() == 3
    nd = eps_u.shape[0] if stacked else 1
    lead = (nd,) if stacked else ()
    S = rff_eps.shape[-2]
    raw_ell = _chk(raw_ell, 'raw_ell', (Do, Di)); raw_var = _chk(raw_var, 'raw_var', (Do,))
    Z = _chk(Z, 'Z', (M, Di)); Um = _chk(Um, 'Um', (M, Do))
    Us_packed = _chk(Us_packed, 'Us_packed', (Do, M * (M + 1) // 2))
    eps_u = _chk(eps_u, 'eps_u', lead + (M, Do))
    rff_w = _chk(rff_w, 'rff_w', lead + (S if kernel == 'RBF' else 2 * S, Do))
    rff_eps = _chk(rff_eps, 'rff_eps', lead + (Di, S, Do)); rff_u = _chk(rff_u, 'rff_u', lead + (1, S, Do))
    pf, wf = cache_sizes(kernel, Di, Do, M, S, nd)
    dev = Z.device
    new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    c = GPCache()
    c.kernel, c.Di, c.Do, c.M, c.S, c.nd, c.stacked = kernel, Di, Do, M, S, nd, stacked
    c.pack, c.ws = new(*lead, pf), new(wf)
    c.ell, c.var, c.omega, c.phase, c.u = new(Do, Di), new(Do), new(*lead, Di, S, Do), new(*lead, 1, S, Do), new(*lead, M, Do)
    c.u_prior = new(*lead, M, Do)
    if kernel == 'RBF':
        c.nu = new(*lead, Do, M, 1)
        c.Lu = new(Do, M, M) if want_Lu else None
    else:
        c.nu = new(*lead, M * Do, 1)
        c.Lu = new(M * Do, M * Do) if want_Lu else None
    _lib.call('gpode_cache_build_fwd_n', KERNEL_ID[kernel], Di, Do, M, S, nd,
              _ptr(raw_ell), _ptr(raw_var), _ptr(Z), _ptr(Um), _ptr(Us_packed),
              _ptr(eps_u), _ptr(rff_w), _ptr(rff_eps), _ptr(rff_u),
              _ptr(c.pack), _ptr(c.ws), _ptr(c.ell), _ptr(c.var), _ptr(c.omega), _ptr(c.phase), _ptr(c.u),
              _ptr(c.Lu), _ptr(c.nu), _ptr(c.u_prior), _stream())
    return c


def _one_draw(cache, what):
    if cache.stacked:
        raise _lib.GpodeError('%s evaluates ONE function draw; this cache holds %d (use rollout / the flow)' % (what, cache.nd))


def kern_cache(kernel, raw_ell, raw_var, rff_w, rff_eps, rff_u):
    """kern.build_cache(S, device) (kernels.py:126-137 / :305-316): fixes the Fourier features of ONE prior draw.  Returns a GPCache
    without inducing records (M = 0): rhs(cache, x, mode=1) is kern.rff_forward(x, S); cache.omega / cache.phase are the
    attributes the reference sets (kern.sample_freq's omega = eps / ell^T inside)."""
    Do, Di = raw_ell.shape
    S = rff_eps.shape[1]
    raw_ell = _chk(raw_ell, 'raw_ell', (Do, Di)); raw_var = _chk(raw_var, 'raw_var', (Do,))
    rff_w = _chk(rff_w, 'rff_w', (S if kernel == 'RBF' else 2 * S, Do))
    rff_eps = _chk(rff_eps, 'rff_eps', (Di, S, Do)); rff_u = _chk(rff_u, 'rff_u', (1, S, Do))
    n = _lib.load().gpode_kern_scratch(KERNEL_ID[kernel], Di, Do, 0, S)
    if n == 0:
        raise _lib.GpodeError('kern.build_cache: no specialisation for kernel=%s D_in=%d D_out=%d' % (kernel, Di, Do))
    new = lambda *s: torch.empty(s, dtype=torch.float32, device=raw_ell.device)
    c = GPCache()
    c.kernel, c.Di, c.Do, c.M, c.S, c.nd, c.stacked = kernel, Di, Do, 0, S, 1, False
    c.pack, c.ws, c.omega, c.phase = new(n), None, new(Di, S, Do), new(1, S, Do)
    c.ell = c.var = c.u = c.Lu = c.nu = c.u_prior = None
    c.noise = dict(rff_w=rff_w, rff_eps=rff_eps, rff_u=rff_u)
    _lib.call('gpode_kern_cache', KERNEL_ID[kernel], Di, Do, S, _ptr(raw_ell), _ptr(raw_var), _ptr(rff_w), _ptr(rff_eps), _ptr(rff_u),
              _ptr(c.pack), _ptr(c.omega), _ptr(c.phase), _stream())
    return c


def compute_nu(kernel, Di, Do, Ku, u_prior, u):
    """kern.compute_nu(Ku, u_prior, inducing_val) (kernels.py:155-172 / :376-387) on the CALLER's kernel matrix: nu = L^-T (u - L^-1 u_prior),
    L = chol(Ku + 1e-5 I).  RBF: Ku (Do,M,M) -> nu (Do,M,1); DF: Ku (M D, M D) -> nu (M D, 1).  Returns (nu, workspace) -- the
    workspace holds the factorisation status (GPCache.check_factorisation reads it the same way)."""
    M = u.shape[0]
    Ku = _chk(Ku, 'Ku', (Do, M, M) if kernel == 'RBF' else (M * Do, M * Do))
    u_prior = _chk(u_prior, 'u_prior', (M, Do)); u = _chk(u, 'inducing_val', (M, Do))
    wf = ctypes.c_size_t(0)
    _lib.call('gpode_compute_nu_ws', KERNEL_ID[kernel], Di, Do, M, ctypes.byref(wf))
    ws = torch.empty(wf.value, dtype=torch.float32, device=u.device)
    nu = torch.empty((Do, M, 1) if kernel == 'RBF' else (M * Do, 1), dtype=torch.float32, device=u.device)
    _lib.call('gpode_compute_nu', KERNEL_ID[kernel], Di, Do, M, _ptr(Ku), _ptr(u_prior), _ptr(u), _ptr(nu), _ptr(ws), _stream())
    return nu, ws


def f_update(kernel, raw_ell, raw_var, x, x2, nu):
    """kern.f_update(x, x2) (kernels.py:174-181 / :390-393): K(x, x2) nu for a nu from compute_nu -> (N, D_out)."""
    Do, Di = raw_ell.shape
    M, N = x2.shape[0], x.shape[0]
    x = _chk(x, 'x', (N, Di)); x2 = _chk(x2, 'x2', (M, Di))
    nu = _chk(nu, 'nu', (Do, M, 1) if kernel == 'RBF' else (M * Do, 1))
    n = _lib.load().gpode_kern_scratch(KERNEL_ID[kernel], Di, Do, M, 0)
    if n == 0:
        raise _lib.GpodeError('kern.f_update: no specialisation for kernel=%s D_in=%d D_out=%d' % (kernel, Di, Do))
    scratch = torch.empty(n, dtype=torch.float32, device=x.device)
    out = torch.empty((N, Do), dtype=torch.float32, device=x.device)
    _lib.call('gpode_f_update', KERNEL_ID[kernel], Di, Do, M, _ptr(_chk(raw_ell, 'raw_ell', (Do, Di))), _ptr(_chk(raw_var, 'raw_var', (Do,))),
              _ptr(x2), _ptr(nu), _ptr(x), N, _ptr(out), _ptr(scratch), _stream())
    return out


def rhs(cache, x, mode=0):
    """SVGP_Layer.forward (svpy.py:123-142): x (N,Di) -> f (N,Do). mode 1: prior only, 2: update only."""
    x = _chk(x, 'x')
    _one_draw(cache, 'rhs')
    if x.dim() != 2 or x.shape[1] != cache.Di:
        raise _lib.GpodeError('x must be (N,%d), got %s' % (cache.Di, tuple(x.shape)))
    N = x.shape[0]
    f = torch.empty((N, cache.Do), dtype=torch.float32, device=x.device)
    _lib.call('gpode_rhs_fwd', KERNEL_ID[cache.kernel], cache.Di, cache.Do, cache.M, cache.S, _ptr(cache.pack),
              _ptr(x), N, _ptr(f), mode, _stream())
    return f


NSTAGE = {'euler': 1, 'rk4': 4, 'midpoint': 2}


def rollout(cache, z0, ts, order, method, save_stages=False):
    """Flow.forward (flow.py:68-86) for a built cache: z0 (N,D), ts (T,) -> zt (N,T,D); a cache of L draws integrates all L * N
    trajectories in ONE launch -> zt (L,N,T,D) (the stack of odegpvae.py:41-44).
    save_stages=True also returns the inputs of all RHS evaluations ([L,] N,T-1,NS,D) for the reverse sweep."""
    if method not in METHOD_ID:
        raise _lib.GpodeError("solver '%s' is not a fixed-grid method of this build (euler, rk4, midpoint)" % method)
    z0 = _chk(z0, 'z0'); ts = _chk(ts, 'ts')
    N, D = z0.shape
    if D != cache.Di or D != order * cache.Do:
        raise _lib.GpodeError('state dim %d must equal D_in=%d = order*D_out=%d' % (D, cache.Di, order * cache.Do))
    T = ts.shape[0]
    lead = cache.lead
    zt = torch.empty(lead + (N, T, D), dtype=torch.float32, device=z0.device)
    xs = torch.empty(lead + (N, max(T - 1, 0), NSTAGE[method], D), dtype=torch.float32, device=z0.device) if save_stages else None
    _lib.call('gpode_rollout_fwd_n', KERNEL_ID[cache.kernel], order, METHOD_ID[method], cache.Di, cache.Do, cache.M,
              cache.S, cache.nd, _ptr(cache.pack), _ptr(z0), _ptr(ts), N, T, _ptr(zt), _ptr(xs), _stream())
    return (zt, xs) if save_stages else zt


def rollout_bwd(cache, xstage, gzt, ts, order, method):
    """Reverse sweep: gzt ([L,] N,T,D) -> gz0 ([L,] N,D), astage ([L,] N,T-1,NS,Do)."""
    gzt = _chk(gzt, 'gzt'); xstage = _chk(xstage, 'xstage'); ts = _chk(ts, 'ts')
    lead = cache.lead
    if gzt.dim() != 3 + len(lead) or tuple(gzt.shape[:len(lead)]) != lead:
        raise _lib.GpodeError('gzt: expected %s + (N,T,D), got %s' % (lead, tuple(gzt.shape)))
    N, T, D = gzt.shape[-3:]
    gz0 = torch.empty(lead + (N, D), dtype=torch.float32, device=gzt.device)
    ast = torch.empty(lead + (N, T - 1, NSTAGE[method], cache.Do), dtype=torch.float32, device=gzt.device)
    _lib.call('gpode_rollout_bwd_n', KERNEL_ID[cache.kernel], order, METHOD_ID[method], cache.Di, cache.Do, cache.M,
              cache.S, cache.nd, _ptr(cache.pack), _ptr(xstage), _ptr(gzt), _ptr(ts), N, T, _ptr(gz0), _ptr(ast), _stream())
    return gz0, ast


# The fused form is OFF by default (GPODE_FUSED_PGRAD=1 switches it on): measured at configs[0] it LENGTHENS the step, 0.737 -> 0.757 ms
# -- the sweep is a latency-bound chain of one workgroup per trajectory, and the parameter terms (376 registers at q = 6) add more to
# every one of its 60 dependent rows than the parameter-sum kernel costs on the side branch; configs[3] (256 trajectories): no difference.
_FUSED_PGRAD = os.environ.get('GPODE_FUSED_PGRAD', '0') == '1'


def pgrad_chunks(cache, N, order, method, force=False):
    """Chunk slabs the fused reverse sweep + parameter sums would use for N trajectories; 0: this shape has no fused form (or the
    fused form is switched off and ``force`` is not set)."""
    if not (_FUSED_PGRAD or force):
        return 0
    return _lib.load().gpode_rollout_bwd_pgrad_chunks(KERNEL_ID[cache.kernel], order, METHOD_ID[method], cache.Di, cache.Do, cache.M,
                                                      cache.S, int(N))


def rollout_bwd_pgrad(cache, xstage, gzt, ts, order, method, nchunk, keep=None):
    """rollout_bwd() and param_grad() in one pass over the rows (gpode_rollout_bwd_pgrad_n): -> gz0, astage, gpack ([L,] pack_floats)."""
    gzt = _chk(gzt, 'gzt'); xstage = _chk(xstage, 'xstage'); ts = _chk(ts, 'ts')
    lead = cache.lead
    N, T, D = gzt.shape[-3:]
    pf = cache.pack.shape[-1]
    gz0 = torch.empty(lead + (N, D), dtype=torch.float32, device=gzt.device)
    ast = torch.empty(lead + (N, T - 1, NSTAGE[method], cache.Do), dtype=torch.float32, device=gzt.device)
    slab = torch.empty(cache.nd * nchunk * pf, dtype=torch.float32, device=gzt.device)
    gpack = torch.empty(lead + (pf,), dtype=torch.float32, device=gzt.device)
    _lib.call('gpode_rollout_bwd_pgrad_n', KERNEL_ID[cache.kernel], order, METHOD_ID[method], cache.Di, cache.Do, cache.M, cache.S,
              cache.nd, _ptr(cache.pack), _ptr(xstage), _ptr(gzt), _ptr(ts), N, T, _ptr(gz0), _ptr(ast), _ptr(slab), nchunk, _ptr(gpack),
              _stream())
    if keep is not None:
        keep.append(slab)
    return gz0, ast, gpack


def rhs_vjp(cache, x, a):
    """gx = J_f(x)^T a for rows x (R,Di), a (R,Do)."""
    x = _chk(x, 'x'); a = _chk(a, 'a')
    _one_draw(cache, 'rhs_vjp')
    gx = torch.empty_like(x)
    _lib.call('gpode_rhs_vjp', KERNEL_ID[cache.kernel], cache.Di, cache.Do, cache.M, cache.S, _ptr(cache.pack),
              _ptr(x), _ptr(a), x.shape[0], _ptr(gx), _stream())
    return gx


_PGRAD_CHUNKS = int(os.environ.get('GPODE_PGRAD_CHUNKS', '0'))     # 0: by the number of rows (below)


def param_grad(cache, x, a, gpack=None, nchunk=None, keep=None):
    """Gradient of sum_r <a_r, f(x_r)> w.r.t. every field of the pack, in pack layout; rows x ([L,] R,Di), a ([L,] R,Do) -- with L
    draws every draw's rows go through its own pack, gpack is (L, pack_floats).
    ``keep`` (a list): the chunk scratch is appended to it -- a caller that launches on a side stream must hold it until that
    stream has been joined (the caching allocator only knows the stream the block was allocated on)."""
    x = _chk(x, 'x'); a = _chk(a, 'a')
    lead = cache.lead
    if x.dim() != 2 + len(lead) or tuple(x.shape[:len(lead)]) != lead:
        raise _lib.GpodeError('x: expected %s + (R,%d), got %s' % (lead, cache.Di, tuple(x.shape)))
    R = x.shape[-2]
    if nchunk is None:
        # one workgroup per chunk of rows, every chunk a full pack-sized slab for the reduction to read: ~32 rows per chunk, between
        # 64 and 256 chunks (configs[0], 1920 rows: 64 chunks 0.855 ms / 256 chunks 0.867 ms per step; configs[1], 15360 rows: 256)
        nchunk = _PGRAD_CHUNKS if _PGRAD_CHUNKS > 0 else min(256, max(64, R // 32))
    nchunk = max(1, min(nchunk, R))
    pf = cache.pack.shape[-1]
    slab = torch.empty(cache.nd * nchunk * pf, dtype=torch.float32, device=x.device)
    acc = 1 if gpack is not None else 0
    if gpack is None:
        # NOT torch.zeros: a torch-native fill launches on torch's current stream, and under launch_on(side) that is not the
        # stream the chunk reduction writes gpack on -- the fill could land after it (seen as a run-to-run wobble of the GP
        # parameter gradients whenever the side stream was ahead).  accumulate = 0: the reduction writes every entry.
        gpack = torch.empty(lead + (pf,), dtype=torch.float32, device=x.device)
    _lib.call('gpode_param_grad_n', KERNEL_ID[cache.kernel], cache.Di, cache.Do, cache.M, cache.S, cache.nd, _ptr(cache.pack),
              _ptr(x), _ptr(a), R, _ptr(slab), nchunk, _ptr(gpack), acc, _stream())
    if keep is not None:
        keep.append(slab)
    return gpack


def kernel_matrix(kernel, raw_ell, raw_var, X, X2=None):
    """kern.K(X, X2) (kernels.py:98-110 / :289-303)."""
    X = _chk(X, 'X')
    X2 = X if X2 is None else _chk(X2, 'X2')
    Do, Di = raw_ell.shape
    N, M2 = X.shape[0], X2.shape[0]
    shape = (Do, N, M2) if kernel == 'RBF' else (N * Do, M2 * Do)
    out = torch.empty(shape, dtype=torch.float32, device=X.device)
    _lib.call('gpode_kernel_matrix', KERNEL_ID[kernel], Di, Do, _ptr(_chk(raw_ell, 'raw_ell')), _ptr(_chk(raw_var, 'raw_var')),
              _ptr(X), N, _ptr(X2), M2, _ptr(out), _stream())
    return out


# ---------------------------------------------------------------------------------------------
# Latent widths outside the compiled specialisations (experiments/main.py:45,77,79 take any integers): the RBF vector field is
# evaluated at the next compiled width on ZERO-PADDED operands, which reproduces the unpadded arithmetic term for term --
#   extra input dimensions: state, inducing locations and frequency noise 0 there, so every difference and phase gains + 0 * 0;
#   extra output dimensions: rff weights, inducing mean and eps_u 0 there, so f_prior(Z), u, nu and f are exactly 0 and the padded
#   state components stay 0 along the whole trajectory (their own K_uu systems are factored and discarded).
# The padding is a differentiable scatter / slice in torch on (M, D)-sized tensors; the kernels are the compiled ones.
# The divergence-free kernel is NOT padded: its matrix-valued kernel carries the width itself (the (D - 1) of kernels.py:296 and the
# normalisation of B(omega), kernels.py:327-336); it is compiled for every width D = 2 .. 16 instead.
# ---------------------------------------------------------------------------------------------
_RBF_COMPILED = None


def _rbf_compiled():
    global _RBF_COMPILED
    if _RBF_COMPILED is None:
        lib = _lib.load()
        _RBF_COMPILED = [(di, do) for do in range(1, 17) for di in range(1, 17) if lib.gpode_supported(0, di, do)]
    return _RBF_COMPILED


class WidthPad:
    """Embedding of an RBF layer of widths (Di, Do) into the compiled widths (Dip, Dop)."""

    def __init__(self, Di, Do):
        self.Di, self.Do = Di, Do
        second_order = Di == 2 * Do                  # state [s, v] of ODEfunc.second_order (flow.py:33-38)
        cands = [(a, b) for a, b in _rbf_compiled() if b >= Do and (a == 2 * b if second_order else (a >= Di and (a == b) == (Di == Do)))]
        if not cands:
            raise _lib.GpodeError('no compiled RBF width can hold D_in=%d, D_out=%d (largest: 16 x 16)' % (Di, Do))
        self.Dip, self.Dop = min(cands, key=lambda c: (c[0] * c[1], c[0]))
        # position of input dimension i in the padded input: order 2 keeps [s | v] as two halves of the padded state
        self.in_index = list(range(Do)) + [self.Dop + j for j in range(Do)] if second_order else list(range(Di))
        self._idx = {}

    def index(self, device):
        key = str(device)
        if key not in self._idx:
            self._idx[key] = torch.tensor(self.in_index, dtype=torch.long, device=device)
        return self._idx[key]

    def pad_in(self, t, dim):
        """zero-pad input-dimension axis ``dim`` of t from Di to Dip (scatter at in_index)."""
        shape = list(t.shape)
        shape[dim] = self.Dip
        return torch.zeros(shape, dtype=t.dtype, device=t.device).index_copy(dim, self.index(t.device), t)

    def pad_out(self, t, dim, value=0.0):
        shape = list(t.shape)
        shape[dim] = self.Dop - self.Do
        if shape[dim] == 0:
            return t
        return torch.cat((t, torch.full(shape, value, dtype=t.dtype, device=t.device)), dim=dim)

    def params(self, raw_ell, raw_var, Z, Um, Us):
        one = 0.5413248546129181                     # invsoftplus(1.0): lengthscale / variance 1 in the padded slots
        # padded input columns of the real rows: raw 0, i.e. lengthscale softplus(0) = 0.69 -- any positive value does, the differences are 0
        ell = self.pad_out(self.pad_in(raw_ell, 1), 0, one)
        M = Z.shape[0]
        P = M * (M + 1) // 2
        diag = torch.zeros(P, dtype=Us.dtype, device=Us.device)
        diag[torch.tensor([n * (n + 1) // 2 + n for n in range(M)], device=Us.device)] = 1e-3
        Usp = Us if self.Dop == self.Do else torch.cat((Us, diag.expand(self.Dop - self.Do, P)), dim=0)
        return ell, self.pad_out(raw_var, 0, one), self.pad_in(Z, 1), self.pad_out(Um, 1), Usp

    def noise(self, nz):
        return dict(eps_u=self.pad_out(nz['eps_u'], 1), rff_w=self.pad_out(nz['rff_w'], 1),
                    rff_eps=self.pad_out(self.pad_in(nz['rff_eps'], 0), 2), rff_u=self.pad_out(nz['rff_u'], 2))

    def state(self, z):
        return self.pad_in(z, z.dim() - 1)

    def unstate(self, z):
        return z.index_select(z.dim() - 1, self.index(z.device))


def width_pad(kernel, Di, Do):
    """None when (kernel, Di, Do) is compiled; a WidthPad for other RBF widths; an error for other DF widths."""
    if _lib.load().gpode_supported(KERNEL_ID[kernel], Di, Do):
        return None
    if kernel != 'RBF':
        raise _lib.GpodeError('the divergence-free kernel is compiled for D = 2 .. 16 with D_in == D_out; (%d, %d) is not (its kernel '
                              'carries the width itself, so it cannot be evaluated on zero-padded operands)' % (Di, Do))
    return WidthPad(Di, Do)


class _SvgpKL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Um, Us, M):
        Um_c, Us_c = _chk(Um, 'Um'), _chk(Us, 'Us_sqrt.optvar')
        kl = torch.empty((), dtype=torch.float32, device=Um.device)
        _lib.call('gpode_svgp_kl_fwd', M, Um_c.shape[1], _ptr(Um_c), _ptr(Us_c), _ptr(kl), _stream())
        ctx.save_for_backward(Um_c, Us_c)
        ctx.M = M
        return kl

    @staticmethod
    def backward(ctx, g):
        Um, Us = ctx.saved_tensors
        g = g.contiguous().float()
        dUm, dUs = torch.empty_like(Um), torch.empty_like(Us)
        _lib.call('gpode_svgp_kl_bwd', ctx.M, Um.shape[1], _ptr(Um), _ptr(Us), _ptr(g), _ptr(dUm), _ptr(dUs), _stream())
        return dUm, dUs, None


def svgp_kl(Um, Us_packed, M):
    """SVGP_Layer.kl (svpy.py:144-175), differentiable."""
    return _SvgpKL.apply(Um, Us_packed, M)


def conditional(raw_ell, raw_var, Z, Um, Us, x, full_cov=False, us_rank1=False):
    """SVGP_Layer.build_conditional (svpy.py:176-210) for the RBF kernel: mean (N,Do) and the marginal variances (N,Do), or the
    full covariance (N,N,Do) with ``full_cov``.  Forward only (the reference has no caller that differentiates it)."""
    Do, Di = raw_ell.shape
    M, N = Z.shape[0], x.shape[0]
    raw_ell = _chk(raw_ell, 'raw_ell', (Do, Di)); raw_var = _chk(raw_var, 'raw_var', (Do,))
    Z = _chk(Z, 'Z', (M, Di)); Um = _chk(Um, 'Um', (M, Do)); x = _chk(x, 'x', (N, Di))
    Us = _chk(Us, 'Us', (Do, M) if us_rank1 else (Do, M * (M + 1) // 2))     # q_diag: columns s_d (the reference's rank-one Us Us^T)
    wf = ctypes.c_size_t(0)
    _lib.call('gpode_conditional_ws', Di, Do, M, N, ctypes.byref(wf))
    new = lambda *s: torch.empty(s, dtype=torch.float32, device=x.device)
    ws, mean = new(wf.value), new(N, Do)
    var = new(N, N, Do) if full_cov else new(N, Do)
    _lib.call('gpode_conditional', Di, Do, M, N, _ptr(raw_ell), _ptr(raw_var), _ptr(Z), _ptr(Um), _ptr(Us), int(bool(us_rank1)), _ptr(x),
              int(bool(full_cov)), _ptr(mean), _ptr(var), _ptr(ws), _stream())
    return mean, var


class _Flow(torch.autograd.Function):
    """GP function draw(s) + fixed-grid integration (flow.py:68-86), differentiable w.r.t. z0 and the five GP parameter tensors.
    ``draws`` None: one draw, zt (N,T,D).  ``draws`` = L: the L draws of ODEGPVAE.sample_trajectories (odegpvae.py:37-45) in one
    pass -- ONE cache build that factors K_uu once, ONE rollout launch over L * N trajectories, zt (L,N,T,D); the backward is one
    reverse sweep, one parameter-sum launch and one cache backward on the gradients summed over the draws.
    Forward: gpode_cache_build_fwd_n + gpode_rollout_fwd_n.  Backward: gpode_rollout_bwd_n (reverse sweep), gpode_param_grad_n
    (pack-layout parameter gradients), gpode_cache_build_bwd_n."""

    @staticmethod
    def forward(ctx, z0, ts, raw_ell, raw_var, Z, Um, Us, gp, order, method, draws=None):
        cache = gp.take_prebuilt_cache() if hasattr(gp, 'take_prebuilt_cache') else None
        if cache is None:
            cache = gp.build_cache() if draws is None else gp.build_cache(draws=draws)
        if (cache.nd if cache.stacked else None) != draws:
            raise _lib.GpodeError('the prebuilt cache holds %s draws, the flow was asked for %s' % (cache.lead or 'one', draws))
        ctx.params = (raw_ell, raw_var, Z, Um, Us)
        need = any(ctx.needs_input_grad)
        ctx.prepared = getattr(cache, 'prepared', None)      # overlap mode: started behind the cache build already (svpy.prebuild_cache)
        if ctx.prepared is None and _overlap['on'] and any(ctx.needs_input_grad[2:7]):
            # L^-1 for the cache backward depends on the forward factor only: start it now on the side stream, where it
            # runs under the rollout / decoder instead of at the exposed end of the backward pass
            side = fork_side_stream()
            with launch_on(side):
                ctx.prepared = cache_bwd_prepare(cache)
        if need:
            zt, xs = rollout(cache, z0, ts, order, method, save_stages=True)
        else:
            zt, xs = rollout(cache, z0, ts, order, method), None
        ctx.cache, ctx.order, ctx.method = cache, order, method
        ctx.save_for_backward(ts, xs, raw_ell.detach(), raw_var.detach(), Z.detach())
        return zt

    @staticmethod
    def backward(ctx, gzt):
        ts, xs, raw_ell, raw_var, Z = ctx.saved_tensors
        c = ctx.cache
        lead = c.lead
        want_p = any(ctx.needs_input_grad[2:7])
        # the reverse sweep visits every (stage input, adjoint) row: where a fused form exists the rows' parameter-gradient terms are
        # summed on the way, and the side branch below starts with the pack gradient instead of with a pass over the rows
        nch = pgrad_chunks(c, gzt.shape[-3], ctx.order, ctx.method) if want_p else 0
        gpack_f = None
        if nch:
            gz0, ast, gpack_f = rollout_bwd_pgrad(c, xs, gzt.contiguous(), ts, ctx.order, ctx.method, nch)
        else:
            gz0, ast = rollout_bwd(c, xs, gzt.contiguous(), ts, ctx.order, ctx.method)
        if c.stacked:
            gz0 = gz0.sum(0)                         # every draw starts from the same z0 (odegpvae.py:42)
        if not want_p:
            return (gz0,) + (None,) * 10
        leaves = all(p.is_leaf and p.requires_grad for p in ctx.params) and all(ctx.needs_input_grad[2:7])
        if _overlap['on'] and leaves:
            # parameter gradients on the side stream, next to the encoder's backward; join_side_stream() adds them
            side = fork_side_stream()
            _main_marker()
            scratch = []
            kl, add_to = _overlap['kl'], None
            if kl is not None and kl[0][0].data_ptr() == ctx.params[3].data_ptr() and kl[0][1].data_ptr() == ctx.params[4].data_ptr():
                add_to, _overlap['kl'] = kl[1], None  # the KL gradients of (Um, Us): the cache backward adds the flow's to them
            with launch_on(side):
                gpack = gpack_f if gpack_f is not None else param_grad(c, xs.reshape(lead + (-1, c.Di)), ast.reshape(lead + (-1, c.Do)),
                                                                        keep=scratch)
                g = cache_build_bwd(c, raw_ell, raw_var, Z, gpack, prepared=ctx.prepared, add_to=add_to)
            grads = [g['raw_ell'], g['raw_var'], g['Z'], g['Um'], g['Us']]
            # every buffer a side-stream kernel touches stays referenced until join_side_stream(): the allocator would
            # otherwise hand the block to the encoder-backward kernels the main stream launches meanwhile
            _overlap['pending'].append((ctx.params, [gg.view_as(p) for gg, p in zip(grads, ctx.params)],
                                        (g, gpack, xs, ast, c, scratch, raw_ell, raw_var, Z, ctx.prepared)))
            return (gz0,) + (None,) * 10
        if ctx.prepared is not None:
            torch.cuda.current_stream().wait_stream(side_stream())
        gpack = gpack_f if gpack_f is not None else param_grad(c, xs.reshape(lead + (-1, c.Di)), ast.reshape(lead + (-1, c.Do)))
        g = cache_build_bwd(c, raw_ell, raw_var, Z, gpack, prepared=ctx.prepared)
        return (gz0, None, g['raw_ell'], g['raw_var'], g['Z'], g['Um'], g['Us'], None, None, None, None)


def flow(gp, z0, ts, order, method, draws=None):
    """One function draw -> zt (N,T,D); ``draws`` = L -> the L draws of odegpvae.py:41-44 in one pass, zt (L,N,T,D)."""
    k = gp.kern
    raw_ell, raw_var = k.raw_dimwise() if hasattr(k, 'raw_dimwise') else (k.unconstrained_lengthscales, k.unconstrained_variance)
    params = (raw_ell, raw_var, gp.inducing_loc.optvar, gp.Um.optvar, gp.us_packed() if hasattr(gp, 'us_packed') else gp.Us_sqrt.optvar)
    pad = getattr(gp, 'width_pad', None)
    if pad is None:
        return _Flow.apply(z0, ts, *params, gp, order, method, draws)
    # a width outside the compiled list: the compiled kernels on zero-padded operands (see WidthPad); autograd carries the
    # gradients back through the scatter / slice
    zt = _Flow.apply(pad.state(z0), ts, *pad.params(*params), gp, order, method, draws)
    return pad.unstate(zt)


def cache_bwd_prepare(cache):
    """The gradient-independent part of cache_build_bwd (L^-1 of the draw's factor), into a workspace that is returned
    and later handed to cache_build_bwd(..., prepared=ws)."""
    c = cache
    bw = ctypes.c_size_t(0)
    _lib.call('gpode_cache_bwd_sizes_n', KERNEL_ID[c.kernel], c.Di, c.Do, c.M, c.S, c.nd, ctypes.byref(bw))
    bws = torch.empty(bw.value, dtype=torch.float32, device=c.pack.device)
    _lib.call('gpode_cache_bwd_prepare_n', KERNEL_ID[c.kernel], c.Di, c.Do, c.M, c.S, c.nd, _ptr(c.ws), _ptr(bws), _stream())
    return bws


def cache_build_bwd(cache, raw_ell, raw_var, Z, gpack, prepared=None, add_to=None):
    """Pack-layout gradient ([L,] pack_floats) -> gradients of the five raw GP parameter tensors (state_dict layouts), summed
    over the draws of the cache.  ``add_to`` = (gUm, gUs): tensors that already hold a gradient of Um / Us (the KL term's); the
    flow's gradient is added to them in place and they are returned as out['Um'], out['Us']."""
    c = cache
    bw = ctypes.c_size_t(0)
    _lib.call('gpode_cache_bwd_sizes_n', KERNEL_ID[c.kernel], c.Di, c.Do, c.M, c.S, c.nd, ctypes.byref(bw))
    dev = gpack.device
    new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    bws = prepared if prepared is not None else new(bw.value)
    out = dict(raw_ell=new(c.Do, c.Di), raw_var=new(c.Do), Z=new(c.M, c.Di))
    if add_to is None:
        out['Um'], out['Us'] = new(c.M, c.Do), new(c.Do, c.M * (c.M + 1) // 2)
    else:
        out['Um'], out['Us'] = _chk(add_to[0], 'gUm', (c.M, c.Do)), _chk(add_to[1], 'gUs', (c.Do, c.M * (c.M + 1) // 2))
    _lib.call('gpode_cache_build_bwd_n', KERNEL_ID[c.kernel], c.Di, c.Do, c.M, c.S, c.nd,
              _ptr(_chk(raw_ell, 'raw_ell')), _ptr(_chk(raw_var, 'raw_var')), _ptr(_chk(Z, 'Z')),
              _ptr(_chk(c.noise['eps_u'], 'eps_u', c.lead + (c.M, c.Do))),
              _ptr(_chk(c.pack, 'pack')), _ptr(c.ws), _ptr(_chk(gpack, 'gpack', c.lead + (c.pack.shape[-1],))), _ptr(bws),
              _ptr(out['raw_ell']), _ptr(out['raw_var']), _ptr(out['Z']), _ptr(out['Um']), _ptr(out['Us']),
              int(prepared is not None) | (2 if add_to is not None else 0), _stream())
    out['_workspace'] = bws   # referenced by the caller for as long as a side stream may still be writing it
    return out
