"""GPU numerics of every conv-VAE HIP op (forward and backward) against the plain PyTorch op of the same
name evaluated in fp64 -- covers the generic direct kernels and the LDS-tiled specialisations (decnn.4 / decnn.7
geometries), ragged batch sizes included.  Tolerance 2e-5 relative to max (fp32 accumulation over <= 1600 terms)."""
import pytest
import torch
import torch.nn.functional as F

from test_gpu_forward import relerr

pytestmark = pytest.mark.gpu
TOL = 2e-5


def check(f_hip, f_ref, *shapes, seed=0):
    g = torch.Generator().manual_seed(seed)
    xs = [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes]
    y = f_hip(*xs)
    w = torch.randn(y.shape, generator=g).cuda()
    (y * w).sum().backward()
    xs64 = [x.detach().double().cpu().requires_grad_(True) for x in xs]
    y64 = f_ref(*xs64)
    (y64 * w.double().cpu()).sum().backward()
    assert relerr(y, y64) < TOL
    for a, b in zip(xs, xs64):
        assert relerr(a.grad, b.grad) < TOL


@pytest.mark.parametrize('B', [1, 5, 40, 130])
def test_decoder_transposed_convs(B):
    from vae_gp_ode_amd import vae_ops as V
    # decnn.1, decnn.4 (tiled), decnn.7 (tiled), decnn.10
    check(lambda x, w, b: V.conv_transpose2d(x, w, b, 1, 0), lambda x, w, b: F.conv_transpose2d(x, w, b), (B, 32, 4, 4), (32, 64, 3, 3), (64,))
    check(lambda x, w, b: V.conv_transpose2d(x, w, b, 2, 1), lambda x, w, b: F.conv_transpose2d(x, w, b, stride=2, padding=1),
          (B, 64, 6, 6), (64, 32, 5, 5), (32,))
    check(lambda x, w, b: V.conv_transpose2d(x, w, b, 2, 1, 1),
          lambda x, w, b: F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1), (B, 32, 13, 13), (32, 16, 5, 5), (16,))
    check(lambda x, w, b: V.conv_transpose2d(x, w, b, 1, 2), lambda x, w, b: F.conv_transpose2d(x, w, b, padding=2),
          (B, 16, 28, 28), (16, 1, 5, 5), (1,))


@pytest.mark.parametrize('B', [1, 4, 33])
def test_encoder_convs(B):
    from vae_gp_ode_amd import vae_ops as V
    for ci, co, h in ((1, 8, 28), (5, 8, 28), (8, 16, 14), (16, 32, 7)):
        check(lambda x, w, b: V.conv2d(x, w, b, 2, 2), lambda x, w, b: F.conv2d(x, w, b, stride=2, padding=2), (B, ci, h, h), (co, ci, 5, 5), (co,))


@pytest.mark.parametrize('frames', [1, 5])
def test_encoder_first_conv_reads_the_minibatch_slice_in_place(frames):
    """encoder(X[:, 0]) / encoder_v(X[:, 0:5]) (odegpvae.py:55-63): the batch-strided slice of X (N,T,1,28,28) goes to the kernel as it is
    (gpode_conv2d_fwd_bs / _bwd_weight_bs) -- same output and weight / bias gradients as on a contiguous copy, against torch in fp64."""
    from vae_gp_ode_amd import vae_ops as V
    g = torch.Generator().manual_seed(3)
    X = torch.randn(9, 7, 1, 28, 28, generator=g)
    w, b = torch.randn(8, frames, 5, 5, generator=g) * 0.2, torch.randn(8, generator=g) * 0.1
    Xd = X.cuda()
    x = Xd[:, 0] if frames == 1 else torch.squeeze(Xd[:, 0:frames])
    assert not x.is_contiguous() and V._batch_strided(x) == 7 * 784
    a = [t.cuda().requires_grad_(True) for t in (w, b)]
    y = V.conv2d(x, a[0], a[1], 2, 2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy.cuda())
    a64 = [t.double().requires_grad_(True) for t in (w, b)]
    x64 = (X[:, 0] if frames == 1 else torch.squeeze(X[:, 0:frames])).double()
    y64 = F.conv2d(x64, a64[0], a64[1], stride=2, padding=2)
    y64.backward(gy.double())
    assert relerr(y, y64) < TOL and relerr(a[0].grad, a64[0].grad) < TOL and relerr(a[1].grad, a64[1].grad) < TOL
    a2 = [t.cuda().requires_grad_(True) for t in (w, b)]
    y2 = V.conv2d(x.contiguous(), a2[0], a2[1], 2, 2)
    y2.backward(gy.cuda())
    assert torch.equal(y, y2) and torch.equal(a[0].grad, a2[0].grad) and torch.equal(a[1].grad, a2[1].grad)


@pytest.mark.parametrize('shape', [(4, 8, 14, 14), (40, 64, 6, 6), (130, 32, 13, 13), (65, 16, 28, 28), (70, 5, 7, 7), (9, 3, 1, 3)])
def test_batchnorm_train_relu(shape):
    from vae_gp_ode_amd import vae_ops as V
    C = shape[1]
    check(lambda x, g, b: V._BatchNormTrain.apply(x, g, b, None, None, 0.1, 1e-5, 1),
          lambda x, g, b: F.relu(F.batch_norm(x, None, None, g, b, True, 0.1, 1e-5)), shape, (C,), (C,))


@pytest.mark.parametrize('shape,relu', [((4, 8, 14, 14), 1), ((130, 32, 13, 13), 1), ((65, 16, 28, 28), 0)])
def test_batchnorm_eval_mode(shape, relu):
    """module.eval(): running statistics; output and input gradient against torch (frozen affine parameters)."""
    from vae_gp_ode_amd import vae_ops as V
    C = shape[1]
    g = torch.Generator().manual_seed(2)
    x = torch.randn(shape, generator=g) * 1.5 + 0.3
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    rm, rv = torch.randn(C, generator=g) * 0.3, torch.rand(C, generator=g) + 0.4
    gy = torch.randn(shape, generator=g)
    xr = x.double().requires_grad_(True)
    yr = F.batch_norm(xr, rm.double(), rv.double(), gam.double(), bet.double(), False, 0.1, 1e-5)
    yr = F.relu(yr) if relu else yr
    yr.backward(gy.double())
    xg = x.cuda().requires_grad_(True)
    y = V._BatchNormEval.apply(xg, gam.cuda(), bet.cuda(), rm.cuda(), rv.cuda(), 1e-5, relu)
    y.backward(gy.cuda())
    assert relerr(y, yr) < 1e-5 and relerr(xg.grad, xr.grad) < 1e-5


def test_batchnorm_running_statistics():
    from vae_gp_ode_amd import vae_ops as V
    bn = torch.nn.BatchNorm2d(16).cuda()
    ref = torch.nn.BatchNorm2d(16).double()
    x = torch.randn(9, 16, 7, 7).cuda() * 2 + 0.5
    for _ in range(3):
        V.batch_norm_train(x, bn, relu=False)
        ref(x.double().cpu())
    assert relerr(bn.running_mean, ref.running_mean) < 1e-5 and relerr(bn.running_var, ref.running_var) < 1e-5
    assert int(bn.num_batches_tracked) == 3


@pytest.mark.parametrize('B', [3, 256, 1030])   # >= 256 rows take the fan-out kernels of the decoder's fc layer
def test_linear_act_loglik(B):
    from vae_gp_ode_amd import vae_ops as V
    check(V.linear, F.linear, (B, 6), (512, 6), (512,))
    check(V.linear, F.linear, (1024 + B, 6), (512, 6), (512,))      # the decoder's fc at thousands of rows: column-owning weight gradient
    check(V.linear, F.linear, (2048 + B, 8), (256, 8), (256,))
    check(V.linear, F.linear, (B, 12), (192, 12), (192,))
    check(V.linear, F.linear, (B, 512), (12, 512), (12,))
    check(V.linear_relu_in, lambda x, w, b: F.linear(F.relu(x), w, b), (B, 512), (12, 512), (12,))   # ReLU folded into the layer (encoder fc)
    check(V.linear_relu_in, lambda x, w, b: F.linear(F.relu(x), w, b), (B, 64), (12, 64), (12,))     # narrow fan-in: separate ReLU
    check(V.relu, F.relu, (B, 33))
    check(V.sigmoid, torch.sigmoid, (B, 33))
    g = torch.Generator().manual_seed(1)
    X = ((torch.rand(B, 4, 1, 28, 28, generator=g) - 0.1307) / 0.3081).cuda()
    for L in (1, 2):
        z = torch.rand(L, B, 4, 1, 28, 28, generator=g).cuda().clamp(1e-3, 1 - 1e-3).requires_grad_(True)
        ll = V.bernoulli_loglik(X, z)
        rs = V.bernoulli_loglik_rowsum(X, z, L * B)
        (ll.sum() + (rs * torch.arange(1, L * B + 1, device='cuda')).sum()).backward()
        z64 = z.detach().double().cpu().requires_grad_(True)
        XL = X.double().cpu().repeat([L, 1, 1, 1, 1, 1])
        ll64 = torch.log(z64) * XL + torch.log(1 - z64) * (1 - XL)
        rs64 = ll64.reshape(L * B, -1).sum(1)
        (ll64.sum() + (rs64 * torch.arange(1, L * B + 1)).sum()).backward()
        assert relerr(ll, ll64) < TOL and relerr(rs, rs64) < TOL and relerr(z.grad, z64.grad) < TOL


@pytest.mark.parametrize('N,q', [(1, 6), (37, 6), (256, 12)])
def test_elbo_glue_ops(N, q):
    """Fused reparameterisation, KL(q(z0) || N(0,I)) rows and loss algebra against the torch expressions of the reference
    (vae.py:75-78; create_model.py:47-49 via torch.distributions; create_model.py:61-73), values and gradients."""
    from torch.distributions import Normal, kl_divergence
    from vae_gp_ode_amd import vae_ops as V
    g = torch.Generator().manual_seed(5)
    mu, logv, eps = torch.randn(N, q, generator=g), torch.randn(N, q, generator=g) * 0.7 - 1.0, torch.randn(N, q, generator=g)
    L = 3
    lh = -torch.rand(L, N, generator=g) * 500 - 100
    klu, nobs = torch.tensor(12.5), 360.0
    wz = torch.randn(N, q, generator=g)

    def ref(mu, logv, lh, klu):
        z = mu + torch.exp(0.5 * logv) * eps.to(mu)
        klr = kl_divergence(Normal(mu, torch.exp(0.5 * logv)), Normal(torch.zeros(q, dtype=mu.dtype), torch.ones(q, dtype=mu.dtype))).sum(-1)
        lhood, klreg = lh.mean(0).mean(), klr.mean()
        return z, klr, torch.stack([-(lhood * nobs - klreg * nobs - klu), -lhood, klreg, klu])

    a64 = [t.double().requires_grad_(True) for t in (mu, logv, lh, klu)]
    z64, klr64, out64 = ref(*a64)
    ((z64 * wz.double()).sum() + out64[0] + 0.3 * out64[2]).backward()
    a = [t.cuda().requires_grad_(True) for t in (mu, logv, lh, klu)]
    z = V.reparam(a[0], a[1], eps.cuda())
    klr = V.normal_kl_rows(a[0], a[1])
    out = V.elbo_terms(a[2], klr, a[3], nobs)
    ((z * wz.cuda()).sum() + out[0] + 0.3 * out[2]).backward()
    assert relerr(z, z64) < 1e-6 and relerr(klr, klr64) < 1e-5 and relerr(out, out64) < 1e-5
    for x, y in zip(a, a64):
        assert relerr(x.grad, y.grad) < 2e-5


@pytest.mark.parametrize('B', [2, 37, 130, 700, 1300])
@pytest.mark.parametrize('geom', [((64, 6, 6), (64, 32, 5, 5), (2, 1, 0)), ((32, 13, 13), (32, 16, 5, 5), (2, 1, 1)),
                                  ((16, 28, 28), (16, 1, 5, 5), (1, 2, 0))])
def test_fused_batchnorm_relu_conv_transpose(B, geom):
    """One decoder stage, ConvTranspose2d(ReLU(BatchNorm2d(c))) in training mode, with the normalised activation never
    materialised (the transposed convolution applies it while staging its input): output, running statistics and every
    gradient (c, gamma, beta, weight, bias) against torch in fp64."""
    from vae_gp_ode_amd import vae_ops as V
    (C, H, _), wshape, (s, p, op) = geom
    g = torch.Generator().manual_seed(11)
    c = torch.randn(B, C, H, H, generator=g) * 1.3 + 0.2
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    w, b = torch.randn(wshape, generator=g) * 0.05, torch.randn(wshape[1], generator=g) * 0.1
    bn = torch.nn.BatchNorm2d(C).cuda()
    ref = torch.nn.BatchNorm2d(C).double()
    with torch.no_grad():
        bn.weight.copy_(gam); bn.bias.copy_(bet); ref.weight.copy_(gam); ref.bias.copy_(bet)
    a64 = [t.double().requires_grad_(True) for t in (c, w, b)]
    y64 = F.conv_transpose2d(F.relu(ref(a64[0])), a64[1], a64[2], stride=s, padding=p, output_padding=op)
    gy = torch.randn(y64.shape, generator=g)
    y64.backward(gy.double())
    a = [t.cuda().requires_grad_(True) for t in (c, w, b)]
    y = V.bn_relu_conv_transpose2d(a[0], bn, a[1], a[2], s, p, op)
    y.backward(gy.cuda())
    assert relerr(y, y64) < TOL
    for x, x64 in zip(a, a64):
        assert relerr(x.grad, x64.grad) < 5 * TOL
    assert relerr(bn.weight.grad, ref.weight.grad) < 5 * TOL and relerr(bn.bias.grad, ref.bias.grad) < 5 * TOL
    assert relerr(bn.running_mean, ref.running_mean) < 1e-5 and relerr(bn.running_var, ref.running_var) < 1e-5
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.gpu
@pytest.mark.parametrize('B', [2, 37, 700, 1300])
def test_decoder_chain_with_statistics_summed_by_the_producing_convolution(B, monkeypatch):
    """decnn.1 -> BatchNorm -> ReLU -> decnn.4 -> BatchNorm -> ReLU -> decnn.7 -> BatchNorm -> ReLU -> decnn.10 (vae.py:107-120) with
    every BatchNorm's batch statistics summed by the transposed convolution in front of it while it stores its output and
    finalised by that kernel's last workgroup (gpode_convT_fwd_stats; all three producers: the taps-as-columns kernels of
    decnn.1 / decnn.4 and the plane engine of decnn.4 / decnn.7), two training steps (the second one shifts the sums by the
    running mean of the first): outputs, every gradient and the running statistics against torch in fp64, and bit-identical
    results from a second run (the combination order does not depend on which workgroup finishes last)."""
    from vae_gp_ode_amd import vae_ops as V
    monkeypatch.setattr(V, '_FUSED_STATS_MIN_IMAGES', 1)
    g = torch.Generator().manual_seed(5)
    ref = torch.nn.Sequential(torch.nn.ConvTranspose2d(32, 64, 3, 1, 0), torch.nn.BatchNorm2d(64), torch.nn.ReLU(),
                              torch.nn.ConvTranspose2d(64, 32, 5, 2, 1), torch.nn.BatchNorm2d(32), torch.nn.ReLU(),
                              torch.nn.ConvTranspose2d(32, 16, 5, 2, 1, output_padding=1), torch.nn.BatchNorm2d(16), torch.nn.ReLU(),
                              torch.nn.ConvTranspose2d(16, 1, 5, 1, 2)).double()
    with torch.no_grad():
        for m in ref:
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5); m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.3)
            if isinstance(m, torch.nn.ConvTranspose2d):
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.5)      # channel means of the order of the spread
    import copy

    def run():
        d = copy.deepcopy(ref).float().cuda()
        outs = []
        for step in range(2):
            x = (torch.randn(B, 32, 4, 4, generator=torch.Generator().manual_seed(20 + step)) * 0.8).cuda().requires_grad_(True)
            for p in d.parameters():
                p.grad = None
            c = V.conv_transpose2d(x, d[0].weight, d[0].bias, 1, 0, stats_for=d[1])
            assert getattr(c, '_gpode_bnstats', None) is not None or not V._fused_stats   # the producer took the statistics
            c = V.bn_relu_conv_transpose2d(c, d[1], d[3].weight, d[3].bias, 2, 1, stats_for=d[4])
            c = V.bn_relu_conv_transpose2d(c, d[4], d[6].weight, d[6].bias, 2, 1, 1, stats_for=d[7])
            assert getattr(c, '_gpode_bnstats', None) is not None or not V._fused_stats
            y = V.bn_relu_conv_transpose2d(c, d[7], d[9].weight, d[9].bias, 1, 2)
            gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(40 + step)).cuda()
            y.backward(gy)
            outs.append((y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in d.parameters()], [b.clone() for b in d.buffers()]))
        return outs
    a, b = run(), run()
    for (y1, gx1, gp1, bf1), (y2, gx2, gp2, bf2) in zip(a, b):
        assert torch.equal(y1, y2) and torch.equal(gx1, gx2) and all(torch.equal(p, q) for p, q in zip(gp1, gp2))
        assert all(torch.equal(p, q) for p, q in zip(bf1, bf2))
    # Gradients are compared in the L2 norm: among the millions of pre-activations of a large batch a few sit within fp32 round-off
    # of zero, fp32 and fp64 then disagree on their ReLU mask, and each such element changes a handful of gradient entries by O(1)
    # -- invisible in the L2 norm, but the whole of a max-norm error (torch's own fp32 pass shows the same against fp64).
    l2 = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())
    r, r32 = copy.deepcopy(ref), copy.deepcopy(ref).float()
    for step, (y, gx, gps, bufs) in enumerate(a):
        x0 = torch.randn(B, 32, 4, 4, generator=torch.Generator().manual_seed(20 + step)) * 0.8
        gy0 = torch.randn(y.shape, generator=torch.Generator().manual_seed(40 + step))
        x64, x32 = x0.double().requires_grad_(True), x0.clone().requires_grad_(True)
        for p in list(r.parameters()) + list(r32.parameters()):
            p.grad = None
        y64 = r(x64)
        y64.backward(gy0.double())
        r32(x32).backward(gy0)
        # (bound: sums of ~1e6 zero-mean terms in fp32 partial sums; torch's CPU BatchNorm accumulates in double and is no yardstick)
        tol = lambda mine, g64, g32: 2e-3 + 4 * l2(g32, g64)
        assert relerr(y, y64) < TOL and l2(gx, x64.grad) < tol(gx, x64.grad, x32.grad), (step, relerr(y, y64), l2(gx, x64.grad), l2(x32.grad, x64.grad))
        for (n, p64), (_, p32), gp in zip(r.named_parameters(), r32.named_parameters(), gps):
            if n in ('0.bias', '3.bias', '6.bias'):    # a bias in front of a BatchNorm has no gradient: round-off on both sides
                assert float(gp.abs().max()) < 10 * float(p32.grad.abs().max()) + 1e-3 and float(p64.grad.abs().max()) < 1e-9, (step, n)
                continue
            assert l2(gp, p64.grad) < tol(gp, p64.grad, p32.grad), (step, n, l2(gp, p64.grad), l2(p32.grad, p64.grad))
        for (n, b64), bf in zip(r.named_buffers(), bufs):
            assert relerr(bf.double(), b64.double()) < 1e-5, (step, n)


@pytest.mark.gpu
@pytest.mark.parametrize('B', [3, 130, 1100])
def test_last_decoder_stage_fused_backward_matches_the_separate_kernels(B):
    """gpode_dec10_bn_bwd_sums / _apply (decnn.10's input gradient recomputed inside both BatchNorm backward passes) against the
    unfused route (gpode_conv2d_fwd + gpode_bn_bwd) on the same inputs, in both forms of the statistics: this rank alone, and
    'gathered' sums with one rank of weight 1 (the data-parallel form; must give the same numbers)."""
    import ctypes
    from vae_gp_ode_amd import _lib, vae_ops as V
    from vae_gp_ode_amd.ops import _ptr, _stream
    g = torch.Generator().manual_seed(B)
    c = (torch.randn(B, 16, 28, 28, generator=g) * 1.3 + 0.2).cuda()
    gam, bet = (torch.rand(16, generator=g) + 0.5).cuda(), (torch.randn(16, generator=g) * 0.3).cuda()
    w, gy = (torch.randn(16, 1, 5, 5, generator=g) * 0.05).cuda(), torch.randn(B, 1, 28, 28, generator=g).cuda()
    mean = c.mean((0, 2, 3))
    invstd = torch.rsqrt(c.var((0, 2, 3), unbiased=False) + 1e-5)
    # separate kernels
    ga = torch.empty_like(c)
    _lib.call('gpode_conv2d_fwd', _ptr(gy), _ptr(w), _ptr(None), _ptr(ga), B, 1, 28, 28, 16, 5, 1, 2, 28, 28, _stream())
    ref = [torch.empty_like(c)] + [torch.empty(16, device='cuda') for _ in range(3)]
    _lib.call('gpode_bn_bwd', _ptr(c), _ptr(ga), _ptr(gam), _ptr(bet), _ptr(mean), _ptr(invstd), *[_ptr(t) for t in ref], B, 16, 784, 1,
              _ptr(V._bn_scratch(B, 16, c)), _stream())
    # fused, local statistics
    out = V._dec10_bn_bwd(None, c, gy, w, gam, bet, mean, invstd)
    for name, a, b in zip(('gc', 'ggamma', 'gbeta', 'chansum'), out, ref):
        tol = 2e-4 if name == 'chansum' else 2e-5        # the channel sums of gc cancel to ~0: compare against the scale of gc
        scale = ref[0].abs().max() * 784 * B if name == 'chansum' else b.abs().max()
        assert float((a - b).abs().max() / scale) < tol, (name, float((a - b).abs().max()), float(scale))
    # fused, gathered statistics of a single rank
    class OneRank:
        world = 1
        def gather(self, sums): return sums.reshape(1, -1).clone()
        def weights(self, device): return torch.ones(1, device=device)
        def count_all(self, n): return float(n)
    out1 = V._dec10_bn_bwd(OneRank(), c, gy, w, gam, bet, mean, invstd)
    for name, a, b in zip(('gc', 'ggamma', 'gbeta'), out1, out):
        assert float((a - b).abs().max() / b.abs().max()) < 1e-6, name
