"""Where the randomness of a GP function draw comes from.

The reference draws inside the modules (kernels.py:13-26,134-137; svpy.py:12-27,94): ``rff_weights``
and the frequency noise from a FRESH UNSEEDED ``np.random.RandomState()`` (not reproducible from
--seed), ``rff_phase`` and the inducing ``epsilon`` from the global numpy RNG.  Here every
draw is an explicit dict of tensors handed to the HIP cache build:

    eps_u (M,Do) ~ N(0,1)      rff_w (S,Do) [DF: (2S,Do)] ~ N(0,1)
    rff_eps (Di,S,Do) ~ N(0,1) rff_u (1,S,Do) ~ U[0,1)

``NumpyNoise`` reproduces the reference's generators and draw order; ``DeviceNoise`` draws on the GPU
(no host round trip; use for throughput) and, under data parallelism, is seeded identically on every
rank so that all shards integrate under the same function draw.
"""
import numpy as np
import torch


def draw_shapes(kernel, Di, Do, M, S, dimwise=True):
    if kernel == 'RBF' and not dimwise:   # shared frequencies (kernels.py:118-119,131)
        return dict(rff_w=(S, Do), rff_eps=(Di, S), rff_u=(1, S), eps_u=(M, Do))
    return dict(rff_w=(S if kernel == 'RBF' else 2 * S, Do), rff_eps=(Di, S, Do), rff_u=(1, S, Do), eps_u=(M, Do))


class NumpyNoise:
    def __init__(self, unseeded_rff=True):
        self.unseeded_rff = unseeded_rff

    def draw(self, kernel, Di, Do, M, S, device, dimwise=True):
        sh = draw_shapes(kernel, Di, Do, M, S, dimwise)
        fresh = np.random.RandomState() if self.unseeded_rff else np.random
        out = dict(rff_w=fresh.normal(size=sh['rff_w']))
        fresh = np.random.RandomState() if self.unseeded_rff else np.random
        out['rff_eps'] = fresh.normal(size=sh['rff_eps'])
        out['rff_u'] = np.random.uniform(low=0.0, high=1.0, size=sh['rff_u'])
        out['eps_u'] = np.random.normal(size=sh['eps_u'])
        return {k: torch.tensor(v.astype(np.float32)).to(device) for k, v in out.items()}


def _device(device):
    dev = torch.device(device)
    return torch.device('cuda', torch.cuda.current_device()) if dev.type == 'cuda' and dev.index is None else dev


class DeviceNoise:
    """Draws on the device with the library's own counter-based generator (gpode_noise_fill: Philox4x32-10 keyed by ``seed``,
    counter = (element, draw number)).  ALL the noise of a step -- the three normal tensors of the function draw(s), the uniform
    phases and, when ``reserve()`` announced them, the encoder's reparameterisation draws (vae.py:76) -- comes out of ONE launch;
    the draw number lives in device memory and is advanced by the kernel, so a step captured into a HIP graph draws fresh numbers at
    every replay with no generator to register and no host-side bookkeeping between replays."""

    def __init__(self, seed=0, device=None):
        self.seed = int(seed)
        self._state = None
        self._device = device
        self._reserved = 0           # normals the next draw appends for normal()
        self._extra = None           # ... and what is left of them

    def state(self, device):
        """{draw number, ticket} in device memory (two 64-bit words), created at draw number 0 on first use."""
        dev = _device(device)
        if self._state is None or self._state.device != dev:
            self._state = torch.zeros(2, dtype=torch.int64, device=dev)
        return self._state

    def manual_seed(self, seed):
        self.seed = int(seed)
        if self._state is not None:
            self._state.zero_()
        return self

    def reserve(self, n):
        """The next draw()/draw_n() also produces ``n`` standard normals for normal() -- the same launch."""
        self._reserved = int(n)
        self._extra = None

    def _fill(self, n_normal, n_uniform, device):
        from ... import _lib, ops
        flat = torch.empty(n_normal + n_uniform, dtype=torch.float32, device=device)
        st = self.state(device)
        _lib.call('gpode_noise_fill', ops._ptr(flat), n_normal, n_uniform, self.seed & 0xFFFFFFFFFFFFFFFF, ops._ptr(st), ops._stream())
        return flat

    def normal(self, shape, device):
        """Standard normals of ``shape``: taken from the block reserve() announced if it is there, else a launch of their own."""
        n = int(np.prod(shape))
        ex = self._extra
        if ex is not None and ex.device == _device(device) and ex.numel() >= n:
            self._extra = ex[n:] if ex.numel() > n else None
            return ex[:n].view(shape)
        return self._fill(n, 0, device).view(shape)

    def _draw(self, sh, lead, device):
        names = ('rff_w', 'rff_eps', 'eps_u')
        sizes = [int(np.prod(lead + tuple(sh[k]))) for k in names]
        sizes = [(n + 3) // 4 * 4 for n in sizes]                  # every tensor starts on a 16-byte boundary
        nu = int(np.prod(lead + tuple(sh['rff_u'])))
        extra, self._reserved = (self._reserved + 3) // 4 * 4, 0
        flat = self._fill(sum(sizes) + extra, nu, device)
        out, o = {}, 0
        for k, n in zip(names, sizes):
            out[k] = flat[o:o + int(np.prod(lead + tuple(sh[k])))].view(lead + tuple(sh[k]))
            o += n
        self._extra = flat[o:o + extra] if extra else None
        out['rff_u'] = flat[o + extra:].view(lead + tuple(sh['rff_u']))
        return out

    def draw(self, kernel, Di, Do, M, S, device, dimwise=True):
        return self._draw(draw_shapes(kernel, Di, Do, M, S, dimwise), (), device)

    def draw_n(self, kernel, Di, Do, M, S, device, L):
        """L draws with a leading draw axis, still one launch: tensor k is the contiguous block [L][shape_k] of the buffer.  (Not
        the same numbers as L successive draw() calls -- the stream is consumed in a different order -- but the same distribution,
        and identical on every rank that seeds alike.)"""
        return self._draw(draw_shapes(kernel, Di, Do, M, S, True), (L,), device)


def install_device_noise(model, seed, eps_seed=None):
    """Device-side randomness for the whole model: the GP layer draws from a DeviceNoise, and the encoders take their
    reparameterisation noise (vae.py:76) from the same source -- one launch per step for all of it.  ``eps_seed`` gives the
    encoders a source of their own instead (data parallelism: every rank integrates under the SAME function draw, but the
    reparameterisation noise of its shard is independent of the other shards')."""
    src = DeviceNoise(seed)
    model.flow.odefunc.diffeq.noise_source = src
    eps = src if eps_seed is None else DeviceNoise(eps_seed)
    for enc in (model.vae.encoder, getattr(model.vae, 'encoder_v', None)):
        if enc is not None:
            enc.eps_source = eps
    return src
