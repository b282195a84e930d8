"""The kernel's own per-draw methods (SURVEY 8b "Methods": kern.build_cache / sample_freq / rff_forward / compute_nu / f_update,
kernels.py:112-181, :305-393) bound one by one, the way a caller that keeps its own SVGP_Layer would use them -- against the values
the reference's methods produced (fixtures captured by tests/golden/make_golden.py from the reference's own modules):
  kern.build_cache(S)            -> rff_omega, rff_phase  (1e-6)            gpode_kern_cache
  kern.rff_forward(x / Z, S)     -> f_prior(x), f_prior(Z) (2e-5)           gpode_rhs_fwd on the prior-only cache
  kern.compute_nu(Ku, u_prior, u)-> nu                    (as test_gpu_forward: 2e-4 + 3 x the reference's own distance to fp64)
  kern.f_update(x, Z)            -> K(x, Z) nu            (same)             gpode_f_update
  kern.sample_freq(S, seed)      -> RandomState(seed).normal / ell^T        (the reference's sample_normal, kernels.py:13-19)"""
import numpy as np
import pytest
import torch

from conftest import load_golden, sub
from oracle import gpode_oracle as O
from test_gpu_forward import relerr, tol_downstream

pytestmark = pytest.mark.gpu

CASES = [('gp_rbf1_tiny', 'RBF', 1, True), ('gp_rbf2_tiny', 'RBF', 2, True), ('gp_df1_tiny', 'DF', 1, True),
         ('gp_df1_tiny_q5', 'DF', 1, True), ('gp_df1_tiny_q10', 'DF', 1, True), ('gp_rbf1_cfg1', 'RBF', 1, True),
         ('gp_df1_cfg2', 'DF', 1, True), ('gp_rbf1_tiny_shared', 'RBF', 1, False)]


def _kern(g, kernel, dimwise):
    from vae_gp_ode_amd.model.core.kernels import RBF, DivergenceFreeKernel
    sd = sub(g, 'sd.flow.odefunc.diffeq.')
    raw_ell, raw_var = sd['kern.unconstrained_lengthscales'], sd['kern.unconstrained_variance']
    Di = raw_ell.shape[-1]
    Do = raw_ell.shape[0] if dimwise else sd['Um.optvar'].shape[1]
    k = (RBF(Di, Do, dimwise) if kernel == 'RBF' else DivergenceFreeKernel(Di, Do)).cuda()
    with torch.no_grad():
        k.unconstrained_lengthscales.copy_(raw_ell)
        k.unconstrained_variance.copy_(raw_var)
    return k, sd


@pytest.mark.parametrize('name,kernel,order,dimwise', CASES, ids=[c[0] for c in CASES])
def test_kernel_methods_match_the_reference_values(name, kernel, order, dimwise):
    g = load_golden(name)
    k, sd = _kern(g, kernel, dimwise)
    nz = sub(g, 'noise.')
    S = nz['rff_eps'].shape[1]
    Z, x = sd['inducing_loc.optvar'].cuda(), g['x'].cuda()
    # kern.build_cache: the Fourier features of the draw
    k.build_cache(S, 'cuda', noise={kk: nz[kk] for kk in ('rff_w', 'rff_eps', 'rff_u')})
    assert tuple(k.rff_omega.shape) == tuple(g['omega'].shape) and tuple(k.rff_phase.shape) == tuple(g['phase'].shape)
    assert relerr(k.rff_omega, g['omega']) < 1e-6 and relerr(k.rff_phase, g['phase']) < 1e-6
    assert torch.equal(k.rff_weights.cpu(), nz['rff_w'])
    # kern.rff_forward on the kernel's own features
    assert relerr(k.rff_forward(x, S), g['f_prior_x']) < 2e-5
    u_prior = k.rff_forward(Z, S)
    assert relerr(u_prior, g['u_prior']) < 2e-5
    # kern.compute_nu from the reference's Ku / u_prior and the inducing sample of the draw
    p = O.gp_params_from_state_dict(sub(g, 'sd.'))
    u = O.sample_inducing(p['Us'], nz['eps_u'], p['Um'])
    nu = k.compute_nu(g['Ku'].cuda(), g['u_prior'].cuda(), u.cuda())
    assert tuple(nu.shape) == tuple(g['nu'].shape)
    if dimwise:
        tol = tol_downstream(name, kernel, order, 'nu', g)
        assert relerr(nu, g['nu']) < tol, (relerr(nu, g['nu']), tol)
        # kern.f_update(x, Z) with that nu
        fu = k.f_update(x, Z)
        tol = tol_downstream(name, kernel, order, 'f_update_x', g)
        assert relerr(fu, g['f_update_x']) < tol, (relerr(fu, g['f_update_x']), tol)
    else:
        assert relerr(nu, g['nu']) < 2e-3 and relerr(k.f_update(x, Z), g['f_update_x']) < 2e-3
    # a layer-level draw takes the kernel back (f_update without compute_nu's nu goes through the layer's cache)
    assert k._kern_nu is not None
    # kern.sample_freq(S, seed): the reference's seeded spectral sample
    om = k.sample_freq(S, seed=7, device='cuda')
    shape = (k.D_in, S, k.D_out) if (dimwise or kernel == 'DF') else (k.D_in, S)
    eps = torch.tensor(np.random.RandomState(7).normal(size=shape).astype(np.float32))
    ell = g['ell']
    ref = eps / (ell.T.unsqueeze(1) if (dimwise or kernel == 'DF') else ell.unsqueeze(1))
    assert tuple(om.shape) == shape and relerr(om, ref) < 1e-6


def test_compute_nu_reports_a_matrix_that_is_not_positive_definite():
    """torch.linalg.cholesky raises inside the reference's compute_nu (kernels.py:163); here the status word of the workspace says so."""
    from vae_gp_ode_amd import ops
    import ctypes
    M, D = 8, 2
    Ku = -torch.eye(M).expand(D, M, M).contiguous().cuda()
    nu, ws = ops.compute_nu('RBF', D, D, Ku, torch.zeros(M, D).cuda(), torch.ones(M, D).cuda())
    info = ctypes.c_int(0)
    from vae_gp_ode_amd import _lib
    _lib.call('gpode_cache_info', ops._ptr(ws), ctypes.byref(info), ops._stream())
    assert info.value & 1
