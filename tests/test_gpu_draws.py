"""The Monte-Carlo axis as a batched axis (odegpvae.py:37-45: `for l in range(L)` over whole flow calls; main.py:200 trains half of
all epochs at L = 5).  The `_n` entry points build ONE Cholesky factor of K_uu for all L draws, solve their right-hand sides as a
block, integrate the L * N trajectories in one launch and run one cache backward on the summed adjoint.  Checked here, through the
C ABI, against L separate single-draw passes of the same library (which the other GPU tests pin to the reference fixtures and to
the fp64 oracle):
  forward  -- pack, u, f_prior(Z), nu, trajectories: BIT-EXACT (the shared factor is the per-draw factor, and every draw's
              substitution / rollout executes the same instruction stream on the same values);
  backward -- reverse sweep and pack-layout parameter sums per draw: BIT-EXACT; the five parameter gradients of a loss coupling
              all draws: the per-draw adjoints are summed in FRONT of the Cholesky backward instead of behind it (same terms,
              another summation order through a factor of condition 1e2 .. 1e3), so they are held to the fp64 gradient of the
              oracle: never more than 3x further from it than the sum of L single-draw backwards is (floor 2e-5),
and against the fp64 oracle for a full model step at L = 3."""
import types

import pytest
import torch

from oracle import gpode_oracle as O
from test_gpu_forward import relerr

pytestmark = pytest.mark.gpu

#        name            kernel Di Do  M    S    N  T  L   route through the factor
SHAPES = [('rbf_lds', 'RBF', 6, 6, 100, 256, 8, 6, 5),      # six 128-row systems resident in LDS (BASELINE configs[0], [3])
          ('rbf2_lds', 'RBF', 6, 3, 100, 256, 8, 6, 3),     # second order (configs[2])
          ('rbf_block_edge', 'RBF', 4, 4, 96, 64, 4, 4, 5),  # M = 96: the five rhs rows open a new 32-block of their own
          ('df_tiny', 'DF', 4, 4, 16, 32, 4, 5, 2),          # one 64-row system in LDS
          ('df_chain', 'DF', 6, 6, 100, 256, 8, 6, 5),       # 600 rows: launch chain + k_solve_back (configs[1])
          ('df_big', 'DF', 8, 8, 128, 64, 4, 4, 2),          # 1024 rows: panelled factor, matrix-core updates, panelled solves
          ('rbf_many', 'RBF', 2, 2, 40, 64, 4, 4, 20),      # more draws than one 16-column solve slab / one round of 8 wavefronts
          ('df_w5', 'DF', 5, 5, 20, 32, 4, 4, 3),           # widths outside the original compiled list
          ('df_w11', 'DF', 11, 11, 16, 32, 4, 4, 2)]       # more draws than one 16-column solve slab / one round of 8 wavefronts


def _params(kernel, Di, Do, M, seed):
    g = torch.Generator().manual_seed(seed)
    # DF: one lengthscale and one variance for all entries -- its matrix-valued kernel is only symmetric positive definite then
    return dict(raw_ell=O.invsoftplus(torch.full((Do, Di), 2.0) + (0.3 if kernel == 'RBF' else 0.0) * torch.rand(Do, Di, generator=g)),
                raw_var=O.invsoftplus(torch.ones(Do) + (0.2 if kernel == 'RBF' else 0.0) * torch.rand(Do, generator=g)),
                Z=torch.randn(M, Di, generator=g), Um=0.1 * torch.randn(M, Do, generator=g),
                Us=O.tril_pack(torch.stack([torch.eye(M)] * Do) * 1e-2 + 1e-3 * torch.randn(Do, M, M, generator=g).tril()))


def _noise(kernel, Di, Do, M, S, L, seed):
    g = torch.Generator().manual_seed(seed)
    return dict(eps_u=torch.randn(L, M, Do, generator=g), rff_w=torch.randn(L, S if kernel == 'RBF' else 2 * S, Do, generator=g),
                rff_eps=torch.randn(L, Di, S, Do, generator=g), rff_u=torch.rand(L, 1, S, Do, generator=g))


def _build(ops, kernel, p, nz):
    return ops.cache_build(kernel, p['raw_ell'], p['raw_var'], p['Z'], p['Um'], p['Us'], nz['eps_u'], nz['rff_w'], nz['rff_eps'], nz['rff_u'])


@pytest.mark.parametrize('name,kernel,Di,Do,M,S,N,T,L', SHAPES, ids=[s[0] for s in SHAPES])
def test_batched_draws_equal_separate_draws(name, kernel, Di, Do, M, S, N, T, L):
    from vae_gp_ode_amd import ops
    dev = torch.device('cuda:0')
    order = Di // Do
    p = {k: v.to(dev) for k, v in _params(kernel, Di, Do, M, 3).items()}
    nz = {k: v.to(dev) for k, v in _noise(kernel, Di, Do, M, S, L, 4).items()}
    g = torch.Generator().manual_seed(5)
    z0 = torch.randn(N, Di, generator=g).to(dev)
    ts = (0.1 * torch.arange(T, dtype=torch.float)).to(dev)
    wgt = torch.randn(L, N, T, Di, generator=g).to(dev)            # dL/dzt of a loss that couples all draws

    cb = _build(ops, kernel, p, nz)
    cb.check_factorisation()
    cb.noise = nz
    assert cb.stacked and cb.nd == L and cb.pack.shape[0] == L
    ztb, xsb = ops.rollout(cb, z0, ts, order, 'rk4', save_stages=True)
    gz0b, astb = ops.rollout_bwd(cb, xsb, wgt, ts, order, 'rk4')
    gpb = ops.param_grad(cb, xsb.reshape(L, -1, Di), astb.reshape(L, -1, Do))
    gb = ops.cache_build_bwd(cb, p['raw_ell'], p['raw_var'], p['Z'], gpb.clone())

    acc = None
    for l in range(L):
        nl = {k: v[l].contiguous() for k, v in nz.items()}
        c1 = _build(ops, kernel, p, nl)
        c1.noise = nl
        used = cb.pack.shape[-1] - (-(Do * Di) % 4 if kernel == 'RBF' else -(2 * Do * Do + Do) % 4)   # the tail is padded to a float4
        assert torch.equal(cb.pack[l][:used], c1.pack[:used]), (name, 'draw', l, 'pack')
        for key in ('u', 'u_prior', 'nu', 'omega', 'phase'):
            assert torch.equal(getattr(cb, key)[l], getattr(c1, key)), (name, 'draw', l, key)
        zt1, xs1 = ops.rollout(c1, z0, ts, order, 'rk4', save_stages=True)
        assert torch.equal(ztb[l], zt1) and torch.equal(xsb[l], xs1), (name, 'trajectories of draw', l)
        gz01, ast1 = ops.rollout_bwd(c1, xs1, wgt[l].contiguous(), ts, order, 'rk4')
        assert torch.equal(gz0b[l], gz01) and torch.equal(astb[l], ast1), (name, 'reverse sweep of draw', l)
        gp1 = ops.param_grad(c1, xs1.reshape(-1, Di), ast1.reshape(-1, Do))
        assert torch.equal(gpb[l][:used], gp1[:used]), (name, 'pack-layout parameter sums of draw', l)
        g1 = ops.cache_build_bwd(c1, p['raw_ell'], p['raw_var'], p['Z'], gp1)
        acc = {k: g1[k].double() if acc is None else acc[k] + g1[k].double() for k in ('raw_ell', 'raw_var', 'Z', 'Um', 'Us')}
    # the fp64 gradient of the same loss (oracle + autograd): K_uu's condition number (1e4 .. 1e5 here) sets how far ANY fp32
    # evaluation sits from it, so the batched backward is held to the distance the sum of single-draw backwards has
    p64 = {k: v.detach().cpu().double().requires_grad_(True) for k, v in p.items()}
    loss = 0.0
    for l in range(L):
        c64 = O.build_cache(p64, {k: v[l].cpu().double() for k, v in nz.items()}, kernel)
        loss = loss + (wgt[l].cpu().double() * O.flow_forward(z0.cpu().double(), ts.cpu().double(), c64, order, 'rk4')).sum()
    loss.backward()
    rep, bad = {}, []
    for k in acc:
        e_b, e_s = relerr(gb[k], p64[k].grad), relerr(acc[k], p64[k].grad)
        rep[k] = '%.0e/%.0e' % (e_b, e_s)
        if not e_b < max(3 * e_s, 2e-5):
            bad.append((k, e_b, e_s))
    print(name, 'L=%d: forward bit-exact; parameter gradients vs fp64, batched / sum of %d single-draw backwards:' % (L, L), rep)
    assert not bad, bad


@pytest.mark.parametrize('kernel,order,overlap', [('RBF', 1, False), ('DF', 1, False), ('RBF', 2, True)])
def test_model_step_with_batched_draws_matches_the_oracle(kernel, order, overlap):
    """compute_loss(model, X, L = 3) + backward through the host mirror (ODEGPVAE.sample_trajectories -> Flow.forward(draws=3)) vs the
    fp64 oracle looping over the three draws as the reference does; with the side-stream overlap on for one case."""
    from vae_gp_ode_amd import ops
    from vae_gp_ode_amd.model.create_model import build_model, compute_loss
    from vae_gp_ode_amd.model.core.initialization import initialize_and_fix_kernel_parameters
    from vae_gp_ode_amd.model.misc.torch_utils import seed_everything
    seed_everything(11)
    q, M, S, N, T, L = (3 if order == 2 else 6), 24, 32, 5, 6, 3
    Di = q * order
    args = types.SimpleNamespace(D_in=Di, D_out=q, num_inducing=M, num_features=S, dimwise=True, q_diag=False, device='cuda', kernel=kernel,
                                 ode=order, solver='rk4', use_adjoint=False, frames=5, n_filt=8, latent_dim=q, Ndata=40, dt=0.1)
    m = build_model(args).cuda()
    initialize_and_fix_kernel_parameters(m, 2.0, 1.0)
    g = torch.Generator().manual_seed(12)
    X = torch.rand(N, T, 1, 28, 28, generator=g)
    nzs = [{k: v[0] for k, v in _noise(kernel, Di, q, M, S, 1, 20 + l).items()} for l in range(L)]
    eps_s = torch.randn(N, q, generator=g)
    eps_v = torch.randn(N, q, generator=g) if order == 2 else None
    sd = {k: (v.detach().cpu().double().clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k and '_num_evals' not in k
              else v.detach().cpu().clone()) for k, v in m.state_dict().items()}
    ref = O.compute_loss(X.double(), sd, [O.to_dtype(nz, torch.float64) for nz in nzs], eps_s.double(), None if eps_v is None else eps_v.double(),
                         kernel=kernel, order=order, method='rk4', dt=0.1, Ndata=40)
    ref['loss'].backward()
    gp = m.flow.odefunc.diffeq
    gp.set_noise(*[{k: v.cuda() for k, v in nz.items()} for nz in nzs])
    m.vae.encoder.next_eps = eps_s.cuda()
    if eps_v is not None:
        m.vae.encoder_v.next_eps = eps_v.cuda()
    ops.set_overlap(overlap)
    try:
        out = compute_loss(m, X.cuda(), L)
        out[0].backward()
        ops.join_side_stream()
    finally:
        ops.set_overlap(False)
    assert gp.cache.stacked and gp.cache.nd == L, 'the L draws did not go through the batched path'
    for got, key in zip(out, ('loss', 'nlhood', 'kl_reg', 'kl_u')):
        e = abs(got.item() - ref[key].item()) / abs(ref[key].item())
        assert e < 5e-5, (key, got.item(), ref[key].item())
    dead_bias = ('cnn.0.bias', 'cnn.3.bias', 'decnn.1.bias', 'decnn.4.bias', 'decnn.7.bias')
    worst = {}
    for k, p in m.named_parameters():
        if k.endswith(dead_bias):
            continue
        worst[k.split('.', 2)[-1]] = relerr(p.grad, sd[k].grad)
    print(kernel, 'order', order, 'L=3 gradients vs fp64 oracle:', {k: '%.0e' % v for k, v in worst.items()})
    assert max(worst.values()) < 2e-3, worst
    # the attributes the reference leaves on `kern` after its loop are those of the LAST draw
    assert torch.equal(gp.kern.rff_weights, nzs[-1]['rff_w'].cuda())


@pytest.mark.parametrize('D,M,S,N,T,L,method', [(6, 100, 256, 32, 16, 1, 'rk4'), (6, 100, 256, 9, 5, 3, 'euler'), (4, 40, 100, 300, 4, 1, 'rk4'),
                                                (3, 20, 64, 7, 6, 2, 'rk4'), (2, 16, 32, 2500, 3, 1, 'euler')])
def test_reverse_sweep_with_parameter_sums_equals_the_two_launches(D, M, S, N, T, L, method):
    """gpode_rollout_bwd_pgrad_n (the reverse sweep accumulating the rows' parameter-gradient terms on the way) against
    gpode_rollout_bwd_n + gpode_param_grad_n: same gz0 and stage adjoints bit for bit, the pack-layout sums to summation order
    (one chunk per trajectory instead of chunks of consecutive rows); shapes without a fused form report 0 chunks."""
    from vae_gp_ode_amd import ops
    dev = torch.device('cuda:0')
    p = {k: v.to(dev) for k, v in _params('RBF', D, D, M, 3).items()}
    nz = {k: (v if L > 1 else v[0]).contiguous().to(dev) for k, v in _noise('RBF', D, D, M, S, L, 4).items()}
    g = torch.Generator().manual_seed(6)
    z0 = torch.randn(N, D, generator=g).to(dev)
    ts = (0.1 * torch.arange(T, dtype=torch.float)).to(dev)
    lead = (L,) if L > 1 else ()
    wgt = torch.randn(lead + (N, T, D), generator=g).to(dev)
    c = _build(ops, 'RBF', p, nz)
    c.noise = nz
    zt, xs = ops.rollout(c, z0, ts, 1, method, save_stages=True)
    gz0, ast = ops.rollout_bwd(c, xs, wgt, ts, 1, method)
    gp = ops.param_grad(c, xs.reshape(lead + (-1, D)), ast.reshape(lead + (-1, D)))
    nch = ops.pgrad_chunks(c, N, 1, method, force=True)
    assert nch == min(N, 2048)
    gz0f, astf, gpf = ops.rollout_bwd_pgrad(c, xs, wgt, ts, 1, method, nch)
    assert torch.equal(gz0, gz0f) and torch.equal(ast, astf)
    used = c.pack.shape[-1] - (-(D * D) % 4)
    assert relerr(gpf[..., :used], gp[..., :used].double()) < 2e-5, relerr(gpf[..., :used], gp[..., :used].double())
    assert ops.pgrad_chunks(c, N, 1, 'midpoint', force=True) == 0                       # no fused form: the caller uses the two launches
    pd = {k: v.to(dev) for k, v in _params('DF', 6, 6, 20, 3).items()}
    nd = {k: v[0].contiguous().to(dev) for k, v in _noise('DF', 6, 6, 20, 32, 1, 4).items()}
    assert ops.pgrad_chunks(_build(ops, 'DF', pd, nd), N, 1, 'rk4', force=True) == 0
