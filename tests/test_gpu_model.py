"""GPU parity of the full path: encoder -> GP-ODE flow -> decoder -> ELBO -> backward, through the mirror of
the reference API, against fixtures captured from the reference's own ODEGPVAE / compute_loss / autograd."""
import types

import pytest
import torch

from conftest import load_golden, sub
from test_gpu_forward import relerr

pytestmark = pytest.mark.gpu

CASES = [('model_rbf1_tiny', dict(), 2), ('model_rbf2_tiny', dict(ode=2, D_in=6, D_out=3, latent_dim=3, solver='euler'), 1),
         ('model_df1_tiny', dict(kernel='DF'), 1)]


def make_model(name, kw, L):
    from vae_gp_ode_amd.model.create_model import build_model
    a = dict(D_in=6, D_out=6, num_inducing=16, num_features=32, dimwise=True, q_diag=False, device='cuda', kernel='RBF',
             ode=1, solver='rk4', use_adjoint=False, frames=5, n_filt=8, latent_dim=6, Ndata=360, dt=0.1)
    a.update(kw)
    g = load_golden(name)
    m = build_model(types.SimpleNamespace(**a)).cuda()
    m.load_state_dict(sub(g, 'sd.'))
    gp = m.flow.odefunc.diffeq
    gp.set_noise(*[{k: v.cuda() for k, v in sub(g, 'noise%d.' % l).items()} for l in range(L)])
    m.vae.encoder.next_eps = g['eps_s'].cuda()
    if a['ode'] == 2:
        m.vae.encoder_v.next_eps = g['eps_v'].cuda()
    return m, g


@pytest.mark.parametrize('name,kw,L', CASES)
def test_forward_matches_reference(name, kw, L):
    m, g = make_model(name, kw, L)
    f = load_golden(name + '_fwd')
    with torch.no_grad():
        Xrec, (s_mu, s_logv), (v_mu, v_logv) = m(g['X'].cuda(), L)
    assert relerr(s_mu, f['s_mu']) < 2e-5 and relerr(s_logv, f['s_logv']) < 2e-5
    if v_mu is not None:
        assert relerr(v_mu, f['v_mu']) < 2e-5 and relerr(v_logv, f['v_logv']) < 2e-5
    assert relerr(Xrec, f['Xrec']) < 2e-4
    # BatchNorm running statistics were updated exactly once, like the reference's forward
    for k, v in sub(g, 'sd_after.').items():
        if 'num_batches' in k:
            continue
        assert relerr(m.state_dict()[k], v) < 1e-4, k


@pytest.mark.parametrize('name,kw,L', CASES)
def test_loss_and_gradients_match_reference(name, kw, L):
    from vae_gp_ode_amd.model.create_model import compute_loss
    m, g = make_model(name, kw, L)
    loss, nlhood, kl_reg, kl_u = compute_loss(m, g['X'].cuda(), L)
    for got, key in ((loss, 'loss'), (nlhood, 'nlhood'), (kl_reg, 'kl_reg'), (kl_u, 'kl_u')):
        assert relerr(got, g[key]) < 1e-4, key
    loss.backward()
    gr = sub(g, 'grad.')
    # fp64 twin of the reference's gradients (oracle), to calibrate the tolerance of the ill-conditioned cases
    from oracle import gpode_oracle as O
    sd64 = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v)
            for k, v in sub(g, 'sd.').items()}
    a = dict(kernel='RBF', ode=1, solver='rk4'); a.update(kw)
    r64 = O.compute_loss(g['X'].double(), sd64, [O.to_dtype(sub(g, 'noise%d.' % l), torch.float64) for l in range(L)],
                         g['eps_s'].double(), g['eps_v'].double() if 'eps_v' in g else None, kernel=a['kernel'],
                         order=a['ode'], method=a['solver'], dt=0.1, Ndata=360)
    r64['loss'].backward()
    dead_bias = {'cnn.0.bias', 'cnn.3.bias', 'decnn.1.bias', 'decnn.4.bias', 'decnn.7.bias'}  # feed a BatchNorm: true gradient 0
    params = dict(m.named_parameters())
    errs, worst = {}, 0.0
    for k, ref in gr.items():
        got = params[k].grad
        if any(k.endswith(d) for d in dead_bias):
            wscale = gr[k[:-4] + 'weight'].abs().max().item()
            assert got.abs().max().item() <= 2e-3 * wscale + 1e-5, k
            continue
        tol = 5e-3 + 3 * relerr(ref, sd64[k].grad)
        errs[k] = relerr(got, ref)
        worst = max(worst, errs[k] / tol)
    print(name, {k.split('.', 2)[-1]: '%.1e' % v for k, v in errs.items() if v > 2e-4})
    # model_rbf1_tiny holds ONE decoder BatchNorm+ReLU output that is 3.1e-7 in the reference (and in fp64) and
    # exactly 0 on the HIP path (fp32 round-off on either side of the ReLU kink; tools/debug_kink.py finds it).
    # The gradient of a piecewise-linear net jumps there, so every gradient of this fixture moves by 0.1-0.7 %;
    # with that unit masked the same way all gradients agree to 1e-6.  Hence the wider bound for this one case.
    base = 1e-2 if name == 'model_rbf1_tiny' else 5e-3
    for k, ref in gr.items():
        if k in errs:
            assert errs[k] < base + 3 * relerr(ref, sd64[k].grad), (k, errs[k])


def test_bias_gradients_ride_on_the_batchnorm_backward():
    """The per-channel sums of a BatchNorm backward's output are the bias gradient of the convolution in front of it;
    they are produced while that output is written and handed to the convolution's backward (no separate pass).
    Every BatchNorm of the model (2 encoder + 3 decoder layers) must deliver one."""
    from vae_gp_ode_amd import vae_ops as V
    from vae_gp_ode_amd.model.create_model import compute_loss
    name, kw, L = CASES[0]
    m, g = make_model(name, kw, L)
    before = V.fused_bias_grads
    loss, *_ = compute_loss(m, g['X'].cuda(), L)
    loss.backward()
    assert V.fused_bias_grads - before == 5


def test_t_custom_extrapolation_prefix_property():
    """ODEGPVAE.forward(X, L, T_custom) (odegpvae.py:51-53, used by the plotting / test-error code to roll out beyond the
    observed window): the first T frames of a longer rollout under the same draw equal the T-frame rollout, and the shape
    follows T_custom."""
    name, kw, L = CASES[2]
    m, g = make_model(name, kw, L)
    X = g['X'].cuda()
    T = X.shape[1]
    gp = m.flow.odefunc.diffeq
    nz = {k: v.cuda() for k, v in sub(g, 'noise0.').items()}
    m.eval()           # evaluation-mode BatchNorm: reconstructions of different lengths then see identical statistics
    with torch.no_grad():
        gp._next_noise.clear(); gp.set_noise(nz); m.vae.encoder.next_eps = g['eps_s'].cuda()
        a, _, _ = m(X, 1)
        gp.set_noise(nz); m.vae.encoder.next_eps = g['eps_s'].cuda()
        b, _, _ = m(X, 1, T_custom=2 * T)
    assert tuple(a.shape) == (1, X.shape[0], T, 1, 28, 28) and tuple(b.shape) == (1, X.shape[0], 2 * T, 1, 28, 28)
    assert torch.equal(a, b[:, :, :T])


@pytest.mark.parametrize('name,kw,L', [CASES[0], CASES[1], CASES[2]])
def test_t_custom_rollout_matches_the_oracle(name, kw, L):
    """ODEGPVAE.forward(X, L, T_custom=2T) in the reference's own mode (training-mode BatchNorm over all N x 2T decoded frames,
    odegpvae.py:51-53 only changes the integration / decoding length) against the pinned oracle's model_forward(T_custom=...)
    in fp64 on the fixture's state, draws and encoder noise."""
    from oracle import gpode_oracle as O
    m, g = make_model(name, kw, L)
    a = dict(kernel='RBF', ode=1, solver='rk4'); a.update(kw)
    T2 = 2 * g['X'].shape[1]
    sd64 = O.to_dtype(sub(g, 'sd.'), torch.float64)
    with torch.no_grad():
        X64, zt64, _, _ = O.model_forward(g['X'].double(), sd64, [O.to_dtype(sub(g, 'noise%d.' % l), torch.float64) for l in range(L)],
                                          g['eps_s'].double(), g['eps_v'].double() if 'eps_v' in g else None, kernel=a['kernel'],
                                          order=a['ode'], method=a['solver'], dt=0.1, T_custom=T2)
        X32, zt32, _, _ = O.model_forward(g['X'], sub(g, 'sd.'), [sub(g, 'noise%d.' % l) for l in range(L)],
                                          g['eps_s'], g['eps_v'] if 'eps_v' in g else None, kernel=a['kernel'],
                                          order=a['ode'], method=a['solver'], dt=0.1, T_custom=T2)
        Xrec, _, _ = m(g['X'].cuda(), L, T_custom=T2)
    assert tuple(Xrec.shape) == (L, g['X'].shape[0], T2, 1, 28, 28)
    e, e32 = relerr(Xrec, X64), relerr(X32, X64)
    print(name, 'T_custom=%d: reconstructions %.1e from fp64 (fp32 oracle: %.1e)' % (T2, e, e32))
    assert e < 2e-4 + 3 * e32


def test_df_kernel_matrix_matches_reference():
    """`gpode_kernel_matrix(kernel=DF)` (DivergenceFreeKernel.K, kernels.py:289-303) through the public kern.K API against the
    fixture's K(Z) (96 x 96, before the jitter) and K(Z, x) (96 x 24, the asymmetric DF block layout of SURVEY F7/F8)."""
    from test_gpu_backward import make_layer
    for name in ('gp_df1_tiny', 'gp_df1_tiny_q4', 'gp_df1_tiny_q5', 'gp_df1_tiny_q10', 'gp_df1_cfg2'):
        g = load_golden(name)
        flow, gp = make_layer(g, 'DF', 1, 'rk4')
        Z = gp.inducing_loc.optvar.detach()
        Ku, Kzx = gp.kern.K(Z), gp.kern.K(Z, g['x'].cuda())
        assert tuple(Ku.shape) == tuple(g['Ku'].shape) and tuple(Kzx.shape) == tuple(g['Kzx'].shape)
        assert relerr(Ku, g['Ku']) < 1e-5 and relerr(Kzx, g['Kzx']) < 1e-5, name


@pytest.mark.parametrize('name,kw,L', CASES)
def test_fused_loss_stage_equals_the_separate_launches(name, kw, L, monkeypatch):
    """compute_loss through gpode_sigmoid_loglik_fwd + gpode_elbo_all_fwd (default) against the same step through the separate
    sigmoid / row-sum / KL / loss launches: the four terms and every parameter gradient."""
    from vae_gp_ode_amd.model import create_model as C
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(C, '_FUSED_LOSS', fused)
        m, g = make_model(name, kw, L)
        out = C.compute_loss(m, g['X'].cuda(), L)
        out[0].backward()
        res[fused] = ([float(t) for t in out], {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    for a, b in zip(*[res[f][0] for f in (True, False)]):
        assert abs(a - b) <= 2e-6 * max(1.0, abs(b)), (res[True][0], res[False][0])
    assert res[True][1].keys() == res[False][1].keys()
    # a bias in front of a BatchNorm has NO gradient: both paths hold round-off there (1e-4 of sums of ~1e2 terms), not comparable digits
    no_grad = ('cnn.0.bias', 'cnn.3.bias', 'decnn.1.bias', 'decnn.4.bias', 'decnn.7.bias')
    for k, gb in res[False][1].items():
        if k.endswith(no_grad):
            scale = float(res[False][1][k[:-4] + 'weight'].abs().max())      # the layer's weight gradient: the size of the terms summed
            assert float(res[True][1][k].abs().max()) < 1e-3 * scale and float(gb.abs().max()) < 1e-3 * scale, (k, scale)
            continue
        assert float((res[True][1][k] - gb).abs().max()) <= 2e-5 * float(gb.abs().max()) + 1e-7, k
