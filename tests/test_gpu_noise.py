"""Device-side noise (gpode_noise_fill, model/core/noise.py: DeviceNoise): the library's counter-based generator that replaces the
reference's host draws (kernels.py:13-26,134-137; svpy.py:12-27,94; vae.py:76) in throughput runs.  Checked: the distributions
(moments + Kolmogorov-Smirnov against N(0,1) / U[0,1)), reproducibility from (seed, draw number), the draw number advancing by one
per launch (also under HIP-graph replay, where no torch generator is registered), and that a step's draws come out of ONE launch."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fill(n_normal, n_uniform, seed, state):
    from vae_gp_ode_amd import _lib, ops
    out = torch.empty(n_normal + n_uniform, device='cuda')
    _lib.call('gpode_noise_fill', ops._ptr(out), n_normal, n_uniform, seed, ops._ptr(state), ops._stream())
    return out


def test_distributions_and_reproducibility():
    from scipy import stats
    st = torch.zeros(2, dtype=torch.int64, device='cuda')
    n, m = 1_000_003, 500_001                          # not multiples of 4: the tails are partial quads
    a = _fill(n, m, 1234, st)
    assert st.tolist() == [1, 0]
    z, u = a[:n].double().cpu().numpy(), a[n:].double().cpu().numpy()
    assert abs(z.mean()) < 4e-3 and abs(z.var() - 1) < 6e-3 and abs(stats.skew(z)) < 1e-2 and abs(stats.kurtosis(z)) < 2e-2
    assert stats.kstest(z[:200000], 'norm').pvalue > 1e-3
    assert u.min() >= 0.0 and u.max() < 1.0 and abs(u.mean() - 0.5) < 2e-3 and abs(u.var() - 1 / 12) < 1e-3
    assert stats.kstest(u[:200000], 'uniform').pvalue > 1e-3
    assert abs(np.corrcoef(z[:-1], z[1:])[0, 1]) < 5e-3           # neighbours (the two outputs of a Box-Muller pair among them)
    # same (seed, draw number) -> same numbers, whatever the split into normal / uniform lengths before the element
    st2 = torch.zeros(2, dtype=torch.int64, device='cuda')
    b = _fill(n, m, 1234, st2)
    assert torch.equal(a, b)
    c = _fill(n, m, 1234, st2)                                    # draw number 1
    assert st2.tolist() == [2, 0] and not torch.equal(a[:1000], c[:1000])
    assert abs(np.corrcoef(a[:100000].cpu().numpy(), c[:100000].cpu().numpy())[0, 1]) < 1e-2
    st3 = torch.zeros(2, dtype=torch.int64, device='cuda')
    d = _fill(n, m, 1235, st3)                                    # another seed
    assert abs(np.corrcoef(a[:100000].cpu().numpy(), d[:100000].cpu().numpy())[0, 1]) < 1e-2
    with pytest.raises(Exception):
        _fill(-1, 0, 1, st)


def test_device_noise_draws_a_step_in_one_launch():
    from vae_gp_ode_amd.model.core.noise import DeviceNoise, draw_shapes
    src = DeviceNoise(7)
    sh = draw_shapes('DF', 6, 6, 100, 256)
    src.reserve(32 * 6)
    nz = src.draw('DF', 6, 6, 100, 256, 'cuda')
    assert {k: tuple(v.shape) for k, v in nz.items()} == {k: tuple(v) for k, v in sh.items()}
    eps = src.normal((32, 6), 'cuda')                  # out of the same launch
    assert src.state('cuda').tolist() == [1, 0] and tuple(eps.shape) == (32, 6)
    assert all(v.is_contiguous() and v.data_ptr() % 16 == 0 for v in nz.values())
    assert 0.0 <= float(nz['rff_u'].min()) and float(nz['rff_u'].max()) < 1.0
    assert abs(float(nz['rff_eps'].mean())) < 0.05 and abs(float(nz['rff_eps'].std()) - 1) < 0.05
    more = src.normal((32, 6), 'cuda')                 # nothing reserved any more: a launch of its own
    assert src.state('cuda').tolist() == [2, 0] and not torch.equal(more, eps)
    stacked = src.draw_n('RBF', 6, 6, 100, 256, 'cuda', 5)
    assert tuple(stacked['eps_u'].shape) == (5, 100, 6) and tuple(stacked['rff_u'].shape) == (5, 1, 256, 6)
    # two sources seeded alike agree (what every rank of a data-parallel run relies on), a reset replays the sequence
    a, b = DeviceNoise(3), DeviceNoise(3)
    da, db = a.draw('RBF', 6, 6, 20, 32, 'cuda'), b.draw('RBF', 6, 6, 20, 32, 'cuda')
    assert all(torch.equal(da[k], db[k]) for k in da)
    a.manual_seed(3)
    again = a.draw('RBF', 6, 6, 20, 32, 'cuda')
    assert all(torch.equal(again[k], da[k]) for k in da)


def test_graph_replay_draws_fresh_numbers_without_a_registered_generator():
    from vae_gp_ode_amd.graph import GraphedStep
    from vae_gp_ode_amd.model.core.noise import DeviceNoise
    src = DeviceNoise(11)
    eager = DeviceNoise(11)

    def step():
        return src.draw('RBF', 6, 6, 20, 32, 'cuda')['eps_u'].clone()
    g = GraphedStep(step, warmup=2)                    # two eager draws (numbers 0 and 1), then the captured launch
    seq = [eager.draw('RBF', 6, 6, 20, 32, 'cuda')['eps_u'].clone() for _ in range(5)]
    for k in range(2, 5):
        out = g()
        torch.cuda.synchronize()
        assert torch.equal(out, seq[k]), k             # replay k continues the eager sequence
    assert src.state('cuda').tolist() == [5, 0]
