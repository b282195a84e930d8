#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules.

Run in the build container only (needs /root/reference; it never travels):

    python tests/golden/make_golden.py

What is genuine reference arithmetic: everything in model.core.{kernels,svpy,
flow(ODEfunc/Flow wrappers),vae,odegpvae}, model.create_model and model.misc.*.
What is NOT: the two third-party imports that are absent here --
  * ``torchsummary`` (no arithmetic; a no-op module object is registered), and
  * ``torchdiffeq`` (fixed-grid stepping).  A module object exposing
    ``odeint``/``odeint_adjoint`` with torchdiffeq's call signature is registered;
    it implements the published fixed-grid Euler and rk4 (= 3/8 rule) schemes with
    one step per output interval.  The integrator is therefore unpinned by
    reference code; the RHS it drives (ODEfunc.forward) is the reference's.
The reference draws its noise from numpy/torch RNGs (partly unseeded);
the three sampler helpers and torch.randn_like are wrapped so every draw is
recorded and stored with the fixture.

``data_case`` pins the data path (data/utils.py, data/mnist.py): ``data.mnist`` imports
``torchvision`` (absent; used only for the MNIST download inside create_rotating_dataset),
so an empty module object of that name is registered for the import and the functions that
never touch it -- Dataset, rot_start, rotate_img, the rotating part of the dataset builder
-- run as the reference wrote them on seeded random images.
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

REF = '/root/reference/experiments'
OUT = os.path.dirname(os.path.abspath(__file__))
warnings.filterwarnings('ignore')


# -- third-party stand-ins -----------------------------------------------------
def _fixed_grid_odeint(func, y0, t, atol=None, rtol=None, method=None, **kw):
    sol = [y0]
    y = y0
    for j in range(len(t) - 1):
        t0, t1 = t[j], t[j + 1]
        dt = t1 - t0
        f0 = func(t0, y)
        if method == 'euler':
            dy = dt * f0
        elif method == 'rk4':
            k1 = f0
            k2 = func(t0 + dt / 3, y + dt * k1 * (1.0 / 3.0))
            k3 = func(t0 + dt * 2 / 3, y + dt * (k2 - k1 * (1.0 / 3.0)))
            k4 = func(t1, y + dt * (k1 - k2 + k3))
            dy = (k1 + 3 * (k2 + k3) + k4) * dt * 0.125
        else:
            raise ValueError(method)
        y = y + dy
        sol.append(y)
    return torch.stack(sol, 0)


def _install_standins():
    ts = types.ModuleType('torchsummary')
    ts.summary = lambda *a, **k: None
    sys.modules['torchsummary'] = ts
    td = types.ModuleType('torchdiffeq')
    td.odeint = _fixed_grid_odeint
    td.odeint_adjoint = _fixed_grid_odeint
    sys.modules['torchdiffeq'] = td


_install_standins()
sys.path.insert(0, REF)
from model.core import kernels as rk, svpy as rs, vae as rv  # noqa: E402
from model import create_model as rcm  # noqa: E402
from model.core.initialization import initialize_and_fix_kernel_parameters  # noqa: E402
from model.misc.constraint_utils import invsoftplus  # noqa: E402


# -- noise recording -----------------------------------------------------------
class Recorder:
    def __init__(self, seed):
        self.rng = np.random.RandomState(seed)
        self.gen = torch.Generator().manual_seed(seed)
        self.log = []

    def normal(self, shape, seed=None):
        v = torch.tensor(self.rng.normal(size=shape).astype(np.float32))
        self.log.append(('normal', v))
        return v

    def uniform(self, shape, seed=None):
        v = torch.tensor(self.rng.uniform(size=shape).astype(np.float32))
        self.log.append(('uniform', v))
        return v

    def randn_like(self, x):
        v = torch.randn(x.shape, generator=self.gen, dtype=x.dtype)
        self.log.append(('randn_like', v))
        return v


def patch(rec):
    rk.sample_normal = rec.normal
    rk.sample_uniform = rec.uniform
    rs.sample_normal = rec.normal
    rs.sample_uniform = rec.uniform
    torch.randn_like = rec.randn_like  # vae.py:77 draws encoder noise through torch.randn_like


def split_gp_noise(log, kernel):
    """Order of draws inside SVGP_Layer.build_cache (svpy.py:103-121):
    kern.build_cache -> rff_weights (normal), rff_eps (normal), rff_u (uniform);
    then sample_inducing -> eps_u (normal)."""
    names = ['rff_w', 'rff_eps', 'rff_u', 'eps_u']
    assert [k for k, _ in log] == ['normal', 'normal', 'uniform', 'normal'], [k for k, _ in log]
    return {n: v for n, (_, v) in zip(names, log)}


class Args:
    pass


def make_args(kernel, order, M, S, q, solver, Ndata=360, q_diag=False, dimwise=True):
    a = Args()
    a.D_in = q * order
    a.D_out = q
    a.num_inducing, a.num_features = M, S
    a.dimwise, a.q_diag, a.device, a.kernel = dimwise, q_diag, 'cpu', kernel
    a.ode, a.solver, a.use_adjoint = order, solver, False
    a.frames, a.n_filt, a.latent_dim, a.Ndata, a.dt = 5, 8, q, Ndata, 0.1
    return a


def build(kernel, order, M, S, q, solver, seed, uniform_hyper, spread=0.25, q_diag=False, dimwise=True):
    np.random.seed(seed)
    torch.manual_seed(seed)
    model = rcm.build_model(make_args(kernel, order, M, S, q, solver, q_diag=q_diag, dimwise=dimwise))
    initialize_and_fix_kernel_parameters(model, lengthscale_value=2.0, variance_value=1.0)
    gp = model.flow.odefunc.diffeq
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        if not uniform_hyper:
            k = gp.kern
            k.unconstrained_lengthscales.data = invsoftplus(
                2.0 * (1 + spread * torch.rand(k.unconstrained_lengthscales.shape, generator=g)))
            k.unconstrained_variance.data = invsoftplus(
                1.0 * (1 + 0.5 * torch.rand(k.unconstrained_variance.shape, generator=g)))
        if q_diag:   # raw softplus scale (M,Do): spread the diagonal over ~1e-3 .. 2e-2
            gp.Us_sqrt.optvar.add_(3.0 * torch.rand(gp.Us_sqrt.optvar.shape, generator=g))
        else:
            gp.Us_sqrt.optvar.add_(0.02 * torch.randn(gp.Us_sqrt.optvar.shape, generator=g))
    return model


def npy(t):
    return t.detach().cpu().numpy()


def gp_case(name, kernel, order, N, M, S, q, T, seed, uniform_hyper=False, spread=0.25, q_diag=False, dimwise=True):
    """GP layer + flow goldens (no images)."""
    out = {}
    Di = q * order
    for solver in ('euler', 'rk4'):
        model = build(kernel, order, M, S, q, solver, seed, uniform_hyper, spread, q_diag, dimwise)
        gp = model.flow.odefunc.diffeq
        rec = Recorder(seed + 7)
        patch(rec)
        g = torch.Generator().manual_seed(seed + 3)
        z0 = torch.randn(N, Di, generator=g).requires_grad_(True)
        x = torch.randn(N, Di, generator=g)
        ts = 0.1 * torch.arange(T, dtype=torch.float)
        zt = model.flow(z0, ts)  # build_cache (draws) + integration
        noise = split_gp_noise(rec.log, kernel)
        k = gp.kern
        if solver == 'euler':
            sd = {kk: npy(v) for kk, v in model.state_dict().items() if kk.startswith('flow.') and 'num_evals' not in kk}
            out.update({'sd.' + kk: v for kk, v in sd.items()})
            out.update({'noise.' + kk: npy(v) for kk, v in noise.items()})
            out['z0'], out['x'], out['ts'] = npy(z0), npy(x), npy(ts)
            Z = gp.inducing_loc()
            Ku = k.K(Z)
            out['ell'], out['var'] = npy(k.lengthscales), npy(k.variance)
            out['omega'], out['phase'] = npy(k.rff_omega), npy(k.rff_phase)
            out['Us_dense'] = npy(gp.Us_sqrt())
            out['Ku'] = npy(Ku)
            out['u_prior'] = npy(k.rff_forward(Z, S))
            out['nu'] = npy(k.nu)
            n = Ku.shape[-1]
            out['Lu'] = npy(torch.linalg.cholesky(Ku + torch.eye(n) * 1e-5))
            out['f_prior_x'] = npy(k.rff_forward(x, S))
            out['f_update_x'] = npy(k.f_update(x, Z))
            out['f_x'] = npy(gp(x))
            out['Kzx'] = npy(k.K(Z, x))
            out['kl_u'] = npy(gp.kl())
        out['zt_' + solver] = npy(zt)
        # gradient of a fixed scalar functional of the trajectory
        gw = torch.randn(zt.shape, generator=torch.Generator().manual_seed(seed + 5))
        if solver == 'euler':
            out['gw'] = npy(gw)
        (zt * gw).sum().backward()
        out['grad_' + solver + '.z0'] = npy(z0.grad)
        for kk, p in model.flow.named_parameters():
            out['grad_' + solver + '.' + kk] = npy(p.grad)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, {k: v.shape for k, v in out.items() if k.startswith(('zt', 'nu', 'Ku'))})


def model_case(name, kernel, order, N, M, S, q, T, L, solver, seed):
    """Full ODEGPVAE forward + compute_loss + backward goldens."""
    out = {}
    model = build(kernel, order, M, S, q, solver, seed, uniform_hyper=False)
    rec = Recorder(seed + 11)
    patch(rec)
    g = torch.Generator().manual_seed(seed + 13)
    X = (torch.rand(N, T, 1, 28, 28, generator=g) - 0.1307) / 0.3081  # data/utils.py:8-15 on synthetic frames
    # Bernoulli targets outside [0,1] are the reference's own behaviour
    sd0 = {kk: npy(v).copy() for kk, v in model.state_dict().items()}
    loss, nlhood, kl_reg, kl_u = rcm.compute_loss(model, X, L)
    loss.backward()
    log = rec.log
    n_enc = 1 if order == 1 else 2
    enc = log[:n_enc]
    assert all(k == 'randn_like' for k, _ in enc)
    out['eps_s'] = npy(enc[0][1])
    if order == 2:
        out['eps_v'] = npy(enc[1][1])
    gp_log = log[n_enc:]
    assert len(gp_log) == 4 * L
    for l in range(L):
        nz = split_gp_noise(gp_log[4 * l:4 * l + 4], kernel)
        out.update({'noise%d.%s' % (l, kk): npy(v) for kk, v in nz.items()})
    out.update({'sd.' + kk: v for kk, v in sd0.items()})
    out['X'] = npy(X)
    out['loss'], out['nlhood'], out['kl_reg'], out['kl_u'] = npy(loss), npy(nlhood), npy(kl_reg), npy(kl_u)
    for kk, p in model.named_parameters():
        out['grad.' + kk] = npy(p.grad)
    # second, noise-replayed forward to capture intermediates (deterministic replay)
    sd1 = model.state_dict()
    for kk in sd1:
        if 'running' in kk or 'num_batches' in kk:
            out['sd_after.' + kk] = npy(sd1[kk])
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, 'loss', float(loss), 'nlhood', float(nlhood), 'kl_reg', float(kl_reg), 'kl_u', float(kl_u))


def model_case_intermediates(name, kernel, order, N, M, S, q, T, L, solver, seed):
    """Same inputs as model_case; stores Xrec / mu / logv / ztL from ODEGPVAE.forward."""
    model = build(kernel, order, M, S, q, solver, seed, uniform_hyper=False)
    rec = Recorder(seed + 11)
    patch(rec)
    g = torch.Generator().manual_seed(seed + 13)
    X = (torch.rand(N, T, 1, 28, 28, generator=g) - 0.1307) / 0.3081
    captured = {}
    orig = model.sample_trajectories

    def spy(z0, T_, L_=1):
        r = orig(z0, T_, L_)
        captured['z0'], captured['ztL'] = z0, r
        return r
    model.sample_trajectories = spy
    with torch.no_grad():
        Xrec, (s_mu, s_logv), (v_mu, v_logv) = model(X, L)
    out = dict(Xrec=npy(Xrec), s_mu=npy(s_mu), s_logv=npy(s_logv), z0=npy(captured['z0']), ztL=npy(captured['ztL']))
    if order == 2:
        out.update(v_mu=npy(v_mu), v_logv=npy(v_logv))
    np.savez_compressed(os.path.join(OUT, name + '_fwd.npz'), **out)


def cond_case(name, order, q, M, N, seed, q_diag=False, dimwise=True):
    """SVGP_Layer.build_conditional (svpy.py:176-210) of the reference, marginal and full covariance, on a layer whose
    variational parameters are moved off their initial values."""
    model = build('RBF', order, M, 32, q, 'euler', seed, False, 0.25, q_diag, dimwise)
    gp = model.flow.odefunc.diffeq
    g = torch.Generator().manual_seed(seed + 5)
    with torch.no_grad():
        gp.Um.optvar.add_(0.3 * torch.randn(gp.Um.optvar.shape, generator=g))
        gp.Us_sqrt.optvar.add_(0.2 * torch.randn(gp.Us_sqrt.optvar.shape, generator=g))
    x = 1.5 * torch.randn(N, q * order, generator=g)
    with torch.no_grad():
        mean, var = gp.build_conditional(x)
        mean_f, cov = gp.build_conditional(x, full_cov=True)
    out = {'sd.' + k: npy(v) for k, v in model.state_dict().items() if k.startswith('flow.odefunc.diffeq')}
    out.update(x=npy(x), mean=npy(mean), var=npy(var), mean_full=npy(mean_f), cov=npy(cov))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)


def data_case(name, seed):
    """Dataset items, rot_start, rotate_img of the reference on seeded random frames."""
    for modname in ('torchvision', 'torchvision.transforms'):
        sys.modules.setdefault(modname, types.ModuleType(modname))
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    from data import utils as rdu, mnist as rdm
    rng = np.random.RandomState(seed)
    seqs = (rng.randint(0, 256, (3, 16, 784)) / 255).astype(np.float32)     # grey levels, like the dataset's
    ds = rdu.Dataset(seqs)
    items = np.stack([npy(ds[i]) for i in range(len(ds))])
    Xtr = torch.tensor(seqs).view(3, 16, 1, 28, 28)
    np.random.seed(seed + 1)
    rot = rdm.rot_start(Xtr, 16, 3)
    digits = (rng.rand(3, 28, 28) * 255).astype(np.uint8)
    angles = np.rad2deg(np.linspace(0, 2 * np.pi, 8)[1:])
    rotated = rdm.rotate_img(torch.tensor(digits), angles)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), seqs=seqs, items=items, rot_seed=np.int64(seed + 1), rot=npy(rot),
                        digits=digits, angles=angles, rotated=rotated)


def _only(names):
    """`python make_golden.py NAME ...` regenerates just those fixtures (every case seeds itself: the files do not depend on
    which others are generated in the same run); without arguments all of them."""
    if not names:
        return
    g = globals()
    for fn in ('gp_case', 'model_case', 'model_case_intermediates', 'cond_case', 'data_case'):
        def gate(name, *a, _f=g[fn], **k):
            if name in names:
                return _f(name, *a, **k)
        g[fn] = gate


if __name__ == '__main__':
    torch.set_num_threads(1)
    _only(set(sys.argv[1:]))
    data_case('data_path', 301)
    # q(f(x)) of the sparse GP layer (build_conditional): dimwise RBF orders 1-2, diagonal inducing covariance, shared hyper-parameters
    cond_case('cond_rbf1', 1, 6, 16, 5, 401)
    cond_case('cond_rbf2', 2, 3, 16, 5, 402)
    cond_case('cond_rbf1_qdiag', 1, 6, 16, 5, 403, q_diag=True)
    cond_case('cond_rbf1_shared', 1, 6, 16, 5, 404, dimwise=False)
    # tiny shapes: every variant
    gp_case('gp_rbf1_tiny', 'RBF', 1, N=4, M=16, S=32, q=6, T=5, seed=101)
    gp_case('gp_rbf2_tiny', 'RBF', 2, N=4, M=16, S=32, q=3, T=5, seed=102)
    gp_case('gp_df1_tiny', 'DF', 1, N=4, M=16, S=32, q=6, T=5, seed=103)
    gp_case('gp_df1_tiny_q4', 'DF', 1, N=5, M=12, S=16, q=4, T=4, seed=104, spread=0.05)
    # latent widths the reference accepts like any other (main.py:45,77,79): an odd one, and one past the register-resident kernels
    gp_case('gp_df1_tiny_q5', 'DF', 1, N=5, M=12, S=16, q=5, T=4, seed=109, spread=0.05)
    gp_case('gp_df1_tiny_q10', 'DF', 1, N=5, M=12, S=16, q=10, T=4, seed=110, spread=0.05)
    # q_diag=True: diagonal inducing covariance, Us_sqrt.optvar (M,Do) under a softplus (svpy.py:79-82,95-96,153-167)
    gp_case('gp_rbf1_tiny_qdiag', 'RBF', 1, N=4, M=16, S=32, q=6, T=5, seed=105, q_diag=True)
    gp_case('gp_df1_tiny_qdiag', 'DF', 1, N=4, M=16, S=32, q=6, T=5, seed=106, q_diag=True)
    # dimwise=False: one lengthscale vector / variance / frequency set shared by all outputs (kernels.py:81-96,108-110,164-181)
    gp_case('gp_rbf1_tiny_shared', 'RBF', 1, N=4, M=16, S=32, q=6, T=5, seed=107, dimwise=False)
    gp_case('gp_rbf2_tiny_shared', 'RBF', 2, N=4, M=16, S=32, q=3, T=5, seed=108, dimwise=False)
    # BASELINE configs cfg1/cfg2/cfg3 at the CPU-runnable batch of 32, README hyper-parameters
    gp_case('gp_rbf1_cfg1', 'RBF', 1, N=32, M=100, S=256, q=6, T=16, seed=121, uniform_hyper=True)
    gp_case('gp_df1_cfg2', 'DF', 1, N=32, M=100, S=256, q=6, T=16, seed=122, uniform_hyper=True)
    gp_case('gp_rbf2_cfg3', 'RBF', 2, N=32, M=100, S=256, q=3, T=16, seed=123, uniform_hyper=True)
    # full model (encoder -> flow -> decoder -> ELBO -> backward)
    for nm, kern, order, q, L, solver, seed in [
            ('model_rbf1_tiny', 'RBF', 1, 6, 2, 'rk4', 201),
            ('model_rbf2_tiny', 'RBF', 2, 3, 1, 'euler', 202),
            ('model_df1_tiny', 'DF', 1, 6, 1, 'rk4', 203)]:
        model_case(nm, kern, order, N=4, M=16, S=32, q=q, T=5, L=L, solver=solver, seed=seed)
        model_case_intermediates(nm, kern, order, N=4, M=16, S=32, q=q, T=5, L=L, solver=solver, seed=seed)
