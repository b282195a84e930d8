"""CPU-only: host-side mirror of the reference API (state_dict layout, transforms, argument surface)."""
import types

import numpy as np
import pytest
import torch

from conftest import load_golden, sub


def make_args(**kw):
    a = dict(D_in=6, D_out=6, num_inducing=16, num_features=32, dimwise=True, q_diag=False, device='cpu', kernel='RBF',
             ode=1, solver='rk4', use_adjoint=False, frames=5, n_filt=8, latent_dim=6, Ndata=360, dt=0.1)
    a.update(kw)
    return types.SimpleNamespace(**a)


@pytest.mark.parametrize('name,kw', [('model_rbf1_tiny', {}), ('model_rbf2_tiny', dict(ode=2, D_in=6, D_out=3, latent_dim=3, solver='euler')),
                                     ('model_df1_tiny', dict(kernel='DF'))])
def test_state_dict_layout_matches_reference(name, kw):
    from vae_gp_ode_amd.model.create_model import build_model
    g = sub(load_golden(name), 'sd.')
    m = build_model(make_args(**kw))
    sd = m.state_dict()
    assert sorted(sd) == sorted(g)
    for k in sd:
        assert tuple(sd[k].shape) == tuple(g[k].shape), k
    m.load_state_dict(g)  # reference checkpoints load unchanged


def test_initial_values_follow_reference_rng_order():
    """SVGP_Layer.__init__ consumes the global numpy RNG exactly like svpy.py:76-86."""
    from vae_gp_ode_amd.model.core.svpy import SVGP_Layer
    np.random.seed(7)
    gp = SVGP_Layer(6, 6, 16, 32)
    np.random.seed(7)
    Z = np.random.normal(size=(16, 6)).astype(np.float32)
    Um = (np.random.normal(size=(16, 6)) * 1e-1).astype(np.float32)
    assert np.array_equal(gp.inducing_loc.optvar.detach().numpy(), Z)
    assert np.array_equal(gp.Um.optvar.detach().numpy(), Um)
    dense = gp.Us_sqrt().detach()
    assert torch.equal(dense, torch.stack([torch.eye(16)] * 6) * 1e-3)


def test_lower_triangular_pack_order_matches_fixture():
    from vae_gp_ode_amd.model.misc.transforms import LowerTriangular
    g = load_golden('gp_rbf1_tiny')
    packed = g['sd.flow.odefunc.diffeq.Us_sqrt.optvar']
    t = LowerTriangular(16, 6)
    assert torch.equal(t.forward_tensor(packed), g['Us_dense'])
    assert torch.equal(t.backward_tensor(g['Us_dense']), packed)
    assert np.array_equal(t.forward(packed.numpy()), g['Us_dense'].numpy())


def test_softplus_roundtrip_and_initialisation():
    from vae_gp_ode_amd.model.create_model import build_model
    from vae_gp_ode_amd.model.core.initialization import initialize_and_fix_kernel_parameters
    from vae_gp_ode_amd.model.misc.constraint_utils import invsoftplus, softplus
    x = torch.tensor([1e-3, 0.2, 2.0, 30.0])
    assert torch.allclose(softplus(invsoftplus(x)), x, rtol=1e-6)
    m = initialize_and_fix_kernel_parameters(build_model(make_args()), 2.0, 1.0, fix=True)
    k = m.flow.odefunc.diffeq.kern
    assert torch.allclose(k.lengthscales, torch.full((6, 6), 2.0), rtol=1e-6)
    assert torch.allclose(k.variance, torch.ones(6), rtol=1e-6)
    assert not k.unconstrained_lengthscales.requires_grad


def test_no_cpu_fallback():
    """The HIP path must fail loudly on CPU tensors instead of silently computing elsewhere."""
    from vae_gp_ode_amd import _lib
    from vae_gp_ode_amd.model.core.svpy import SVGP_Layer
    np.random.seed(0)
    gp = SVGP_Layer(6, 6, 16, 32)
    with pytest.raises(_lib.GpodeError):
        gp.build_cache()


def test_cli_argument_surface_matches_reference():
    """Every flag of experiments/main.py:23-114 with the same default and type behaviour."""
    from vae_gp_ode_amd.main import FLAGS, make_parser
    assert len(FLAGS) == 39
    a = make_parser().parse_args([])
    expect = dict(data_root='data/', task='mnist', mask=True, value=3, data_seqlen=100, batch=20, T=16, Ndata=360, Ntest=40,
                  rotrand=True, latent_dim=6, n_filt=8, frames=5, pretrained=False, kernel='RBF', num_features=256,
                  num_inducing=100, dimwise=True, variance=0.7, lengthscale=2.0, q_diag=False, ode=1, D_in=6, D_out=6,
                  solver='euler', ts_dense_scale=2, use_adjoint=False, dt=0.1, Nepoch=5000, lr=0.001, eval_sample_size=128,
                  save='results/mnist', seed=121, log_freq=5, device='cuda:0', continue_training=False, model_path='None', Troll=2)
    for k, v in expect.items():
        assert getattr(a, k) == v, k
    b = make_parser().parse_args(['--kernel', 'DF', '--solver', 'rk4', '--dimwise', 'False', '--ode', '2', '--mask', 'False'])
    assert b.kernel == 'DF' and b.solver == 'rk4' and b.dimwise is False and b.ode == 2 and b.mask is False
    with pytest.raises(SystemExit):  # 'euler' is only accepted as the default
        make_parser().parse_args(['--solver', 'euler'])
