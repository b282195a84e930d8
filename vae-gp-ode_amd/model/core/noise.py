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


class DeviceNoise:
    def __init__(self, seed=0, device=None):
        self.seed = seed
        self._gen = None
        self._device = device

    def generator(self, device):
        """The device generator (created on first use, WITHOUT consuming a draw): a HIP graph that replays draws must have it
        registered before capture (graph.device_generators)."""
        if self._gen is None or self._gen.device != torch.device(device):
            self._gen = torch.Generator(device=device)
            self._gen.manual_seed(self.seed)
        return self._gen

    def draw(self, kernel, Di, Do, M, S, device, dimwise=True):
        sh = draw_shapes(kernel, Di, Do, M, S, dimwise)
        g = self.generator(device)
        # one launch for all normal draws (the three tensors are contiguous slices of one buffer), one for the uniform phase
        names = ('rff_w', 'rff_eps', 'eps_u')
        sizes = [int(np.prod(sh[k])) for k in names]
        flat = torch.randn(sum(sizes), generator=g, device=device)
        out, o = {}, 0
        for k, n in zip(names, sizes):
            out[k] = flat[o:o + n].view(sh[k])
            o += n
        out['rff_u'] = torch.rand(sh['rff_u'], generator=g, device=device)
        return out

    def draw_n(self, kernel, Di, Do, M, S, device, L):
        """L draws with a leading draw axis, still two launches (one normal, one uniform): tensor k is the contiguous block
        [L][shape_k] of the normal buffer.  (Not the same numbers as L successive draw() calls -- the stream is consumed in a
        different order -- but the same distribution, and identical on every rank that seeds alike.)"""
        sh = draw_shapes(kernel, Di, Do, M, S, True)
        g = self.generator(device)
        names = ('rff_w', 'rff_eps', 'eps_u')
        sizes = [L * int(np.prod(sh[k])) for k in names]
        flat = torch.randn(sum(sizes), generator=g, device=device)
        out, o = {}, 0
        for k, n in zip(names, sizes):
            out[k] = flat[o:o + n].view((L,) + tuple(sh[k]))
            o += n
        out['rff_u'] = torch.rand((L,) + tuple(sh['rff_u']), generator=g, device=device)
        return out
