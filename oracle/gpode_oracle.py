"""CPU oracle for the VAE-GP-ODE hot path.  TEST INFRASTRUCTURE ONLY.

This file is a functional, noise-in restatement (plain torch CPU ops, fp32 or
fp64) of the reference's algorithm for the path named in BASELINE.json.  It is
the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product path
(``vae_gp_ode_amd``) never imports anything from ``oracle/``.

Parity pin: every function below is checked bit-for-bit (fp32) against outputs
of the reference's own modules captured in ``tests/golden/*.npz`` by
``tests/golden/make_golden.py`` (see tests/test_oracle_golden.py).  The one
piece that is NOT pinned by reference code is the fixed-grid integrator:
``torchdiffeq`` is an un-vendored, un-pinned third-party dependency that is
absent from this container, so ``odeint_fixed`` restates its published
fixed-grid Euler / ``rk4`` (3/8-rule) scheme -- "integrator parity unpinned".

All citations are relative to /root/reference/experiments/.

Conventions
-----------
N minibatch, M inducing points, S RFF features, Di/Do GP in/out dims,
q latent dim, T time points.  All randomness enters as explicit tensors:
``eps_u (M,Do)``, ``rff_w (S,Do)`` [DF: (2S,Do)], ``rff_eps (Di,S,Do)``,
``rff_u (1,S,Do)`` in [0,1), ``eps_z (N,q)``.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

JITTER = 1e-5  # model/core/kernels.py:11, model/core/svpy.py:10


# ----------------------------------------------------------------------------
# constrained parameters
# ----------------------------------------------------------------------------
def softplus(x):
    """model/misc/constraint_utils.py:5-7"""
    return F.softplus(x) + 1e-12


def invsoftplus(x):
    """model/misc/constraint_utils.py:10-13"""
    xs = torch.max(x - 1e-12, torch.tensor(torch.finfo(x.dtype).eps).to(x))
    return xs + torch.log(-(torch.exp(-xs) - 1))


def tril_unpack(packed, M):
    """model/misc/transforms.py:71-77 -- packed (Do, M(M+1)/2) in row-major
    np.tril_indices order -> dense lower-triangular (Do, M, M)."""
    Do = packed.shape[0]
    out = torch.zeros((Do, M, M), dtype=packed.dtype)
    r, c = np.tril_indices(M, 0)
    out[:, torch.as_tensor(r), torch.as_tensor(c)] = packed
    return out


def tril_pack(dense):
    """model/misc/transforms.py:67-69"""
    M = dense.shape[-1]
    r, c = np.tril_indices(M)
    return torch.stack([d[torch.as_tensor(r), torch.as_tensor(c)] for d in dense])


# ----------------------------------------------------------------------------
# RBF (dimwise) kernel   model/core/kernels.py:29-195
# ----------------------------------------------------------------------------
def rbf_sqdist(X, X2, ell):
    """kernels.py:64-79: expanded-form scaled distance, no clamp. -> (Do,N,M)"""
    Xs_ = X.unsqueeze(0) / ell.unsqueeze(1)
    Xs = torch.sum(torch.pow(Xs_, 2), dim=2)
    if X2 is None:
        return -2 * torch.einsum('dnk, dmk -> dnm', Xs_, Xs_) + Xs.unsqueeze(-1) + Xs.unsqueeze(1)
    X2_ = X2.unsqueeze(0) / ell.unsqueeze(1)
    X2s = torch.sum(torch.pow(X2_, 2), dim=2)
    return -2 * torch.einsum('dnk, dmk -> dnm', Xs_, X2_) + Xs.unsqueeze(-1) + X2s.unsqueeze(1)


def rbf_K(X, X2, ell, var):
    """kernels.py:98-107 -> (Do,N,M)"""
    return var[:, None, None] * torch.exp(-0.5 * rbf_sqdist(X, X2, ell))


def rff_omega(rff_eps, ell):
    """kernels.py:112-124: omega[i,s,d] = eps[i,s,d] / ell[d,i] (RBF and DF)."""
    return rff_eps / ell.T.unsqueeze(1)


def rff_phase(rff_u):
    """kernels.py:136-137 / :315-316: phase = u * 2 * pi (fp32 order kept)."""
    return rff_u * 2 * np.pi


def rbf_rff_forward(x, omega, phase, w, var, S):
    """kernels.py:140-153 -> (N,Do)"""
    xo = torch.einsum('nd,dfk->nfk', x, omega)
    phi = torch.cos(xo + phase) * torch.sqrt(var / S)
    return torch.einsum('nfk,fk->nk', phi, w)


def rbf_compute_nu(Ku, u_prior, u):
    """kernels.py:155-172 (dimwise branch) -> Lu (Do,M,M), nu (Do,M,1)"""
    Lu = torch.linalg.cholesky(Ku + torch.eye(Ku.shape[-1], dtype=Ku.dtype) * JITTER)
    nu = torch.linalg.solve_triangular(Lu, u_prior.T.unsqueeze(2), upper=False)
    nu = torch.linalg.solve_triangular(Lu.permute(0, 2, 1), u.T.unsqueeze(2) - nu, upper=True)
    return Lu, nu


def rbf_f_update(x, Z, nu, ell, var):
    """kernels.py:174-181 -> (N,Do)"""
    Kuf = rbf_K(Z, x, ell, var)
    return torch.einsum('dm, dmn -> nd', nu.squeeze(2), Kuf)


# ----------------------------------------------------------------------------
# Divergence-free kernel   model/core/kernels.py:201-393
# ----------------------------------------------------------------------------
def df_sqdist(X, X2):
    """kernels.py:217-230: UNSCALED expanded-form distance -> (N,M)"""
    Xs = torch.sum(torch.pow(X, 2), dim=1)
    if X2 is None:
        return -2 * torch.matmul(X, X.t()) + torch.reshape(Xs, (-1, 1)) + torch.reshape(Xs, (1, -1))
    X2s = torch.sum(torch.pow(X2, 2), dim=1)
    return -2 * torch.matmul(X, X2.t()) + torch.reshape(Xs, (-1, 1)) + torch.reshape(X2s, (1, -1))


def df_K(X, X2, ell, var):
    """kernels.py:289-303 -> (N*D, M*D), row (n,a), col (m,b).
    ell (D,D) is indexed by the (a,b) block entry, var (D,) by column b."""
    D = X.shape[1]
    N = X.shape[0]
    M = N if X2 is None else X2.shape[0]
    sq = df_sqdist(X, X2)
    l2 = ell.pow(2)
    rbf_term = var * torch.exp(-(1 / (2 * l2) * sq[:, :, None, None]))
    XX2 = X if X2 is None else X2
    diff = torch.subtract(XX2.T[:, None, :], X.T[:, :, None])  # (D,N,M): x2 - x
    term1 = 1 / l2 * torch.multiply(diff[:, None, :, :], diff[None, :, :, :]).permute((2, 3, 0, 1))
    term2 = ((D - 1.0) - (1 / l2) * sq[:, :, None, None]) * torch.eye(D, dtype=X.dtype)[None, None, :, :]
    K = rbf_term * (term1 + term2) / l2
    return torch.reshape(torch.permute(K, (0, 2, 1, 3)), (N * D, M * D))


def df_B_omega(omega):
    """kernels.py:327-337 -> (2S,D,D); x-independent."""
    D = omega.shape[0]
    o1 = omega.permute(1, 0, 2)
    o2 = omega.permute(1, 2, 0)
    norm = torch.sqrt(omega.pow(2).sum(dim=0))[:, None]
    b = norm * torch.eye(D, dtype=omega.dtype)[None, :] - (o1 @ o2) / norm
    return torch.cat((b, b), 0)


def df_rff_forward(x, omega, phase, w, var, S, B=None):
    """kernels.py:319-351 -> (N,D); w is (2S,D).  B: optional precomputed df_B_omega(omega) (tests that
    differentiate w.r.t. B as an independent quantity pass it in)."""
    if B is None:
        B = df_B_omega(omega)
    xo = torch.einsum('nd,dfk->nfk', x, omega)
    phi_ = torch.cat((torch.cos(xo + phase), torch.sin(xo + phase)), 1).unsqueeze(-1)
    phi = (phi_ * B.unsqueeze(0)) * torch.sqrt(var / S)
    return (phi * w[None, :, :, None]).sum([1, 2])


def df_compute_nu(Ku, u_prior, u):
    """kernels.py:376-387 -> Lu (MD,MD), nu (MD,1).  Ku may be asymmetric (F7):
    cholesky reads the lower triangle only."""
    n = Ku.shape[0]
    Lu = torch.linalg.cholesky(Ku + torch.eye(n, dtype=Ku.dtype) * JITTER)
    nu = torch.linalg.solve_triangular(Lu, u_prior.reshape(n)[:, None], upper=False)
    nu = torch.linalg.solve_triangular(Lu.T, u.reshape(n)[:, None] - nu, upper=True)
    return Lu, nu


def df_f_update(x, Z, nu, ell, var):
    """kernels.py:390-393 -> (N,D)"""
    Kuf = df_K(Z, x, ell, var)
    return torch.einsum('md, mn -> nd', nu, Kuf).reshape(x.shape)


# ----------------------------------------------------------------------------
# SVGP layer   model/core/svpy.py
# ----------------------------------------------------------------------------
def is_q_diag(Us, Um):
    """q_diag=True stores Us_sqrt.optvar as the raw (M,Do) softplus scale (svpy.py:79-82); otherwise (Do, M(M+1)/2)."""
    M, Do = Um.shape
    return tuple(Us.shape) == (M, Do) and (M, Do) != (Do, M * (M + 1) // 2)


def sample_inducing(Us_packed, eps_u, Um):
    """svpy.py:88-101 -> (M,Do); q_diag (svpy.py:95-96): softplus(raw) * eps"""
    M = Um.shape[0]
    if is_q_diag(Us_packed, Um):
        return softplus(Us_packed) * eps_u + Um
    Ls = tril_unpack(Us_packed, M)
    return torch.einsum('dnm, md->nd', Ls, eps_u) + Um


def broadcast_shared(p, noise):
    """dimwise=False RBF (kernels.py:45-46,81-96,108-110,118-124,131-132,164-181): one lengthscale vector (D_in,), one
    variance (1,), one frequency / phase set (D_in,S) / (1,S) shared by every output, one Cholesky of the single (M,M)
    K_uu.  Formula by formula this is the dimwise=True case with the hyper-parameters and frequencies repeated along the
    output axis (the D_out factorisations are then identical), so the restatement repeats them; the fixtures
    gp_rbf*_tiny_shared, captured from the reference with dimwise=False, pin the equivalence."""
    if p['raw_ell'].dim() != 1:
        return p, noise
    Do = p['Um'].shape[1]
    p = dict(p, raw_ell=p['raw_ell'].unsqueeze(0).expand(Do, -1), raw_var=p['raw_var'].expand(Do))
    noise = dict(noise, rff_eps=noise['rff_eps'].unsqueeze(-1).expand(-1, -1, Do), rff_u=noise['rff_u'].unsqueeze(-1).expand(-1, -1, Do))
    return p, noise


def build_cache(p, noise, kernel):
    """svpy.py:103-121.  p: dict(raw_ell, raw_var, Z, Um, Us) of optvars;
    noise: dict(eps_u, rff_w, rff_eps, rff_u).  Returns the per-draw cache."""
    p, noise = broadcast_shared(p, noise)
    ell, var = softplus(p['raw_ell']), softplus(p['raw_var'])
    S = noise['rff_eps'].shape[1]
    omega = rff_omega(noise['rff_eps'], ell)
    phase = rff_phase(noise['rff_u'])
    w = noise['rff_w']
    u = sample_inducing(p['Us'], noise['eps_u'], p['Um'])
    Z = p['Z']
    if kernel == 'RBF':
        Ku = rbf_K(Z, None, ell, var)
        u_prior = rbf_rff_forward(Z, omega, phase, w, var, S)
        Lu, nu = rbf_compute_nu(Ku, u_prior, u)
    elif kernel == 'DF':
        Ku = df_K(Z, None, ell, var)
        u_prior = df_rff_forward(Z, omega, phase, w, var, S)
        Lu, nu = df_compute_nu(Ku, u_prior, u)
    else:
        raise ValueError(kernel)
    return dict(kernel=kernel, ell=ell, var=var, S=S, omega=omega, phase=phase, w=w, Z=Z,
                u=u, Ku=Ku, u_prior=u_prior, Lu=Lu, nu=nu)


def gp_prior(x, c):
    if c['kernel'] == 'RBF':
        return rbf_rff_forward(x, c['omega'], c['phase'], c['w'], c['var'], c['S'])
    return df_rff_forward(x, c['omega'], c['phase'], c['w'], c['var'], c['S'], c.get('B'))


def gp_update(x, c):
    fn = rbf_f_update if c['kernel'] == 'RBF' else df_f_update
    return fn(x, c['Z'], c['nu'], c['ell'], c['var'])


def gp_forward(x, c):
    """svpy.py:123-142: f(x) = f_prior(x) + K(x,Z) nu"""
    return gp_prior(x, c) + gp_update(x, c)


def build_conditional(p, x, full_cov=False):
    """svpy.py:176-210 (RBF): q(f(x)) = N(m, Sigma), A = L^-1 K(Z,x), m = A^T Um, Sigma = K(x,x) + A^T (Us Us^T - I) A.
    Returns (mean (N,Do), var (N,Do)) or var (N,N,Do) with full_cov, the layouts of the reference's ``var.T``."""
    p, _ = broadcast_shared(p, dict(rff_eps=torch.zeros(1, 1), rff_u=torch.zeros(1, 1)))
    ell, var = softplus(p['raw_ell']) + 1e-12, softplus(p['raw_var']) + 1e-12          # kernels.py:56-62
    Z, Um = p['Z'], p['Um']
    M = Z.shape[0]
    Ku = rbf_K(Z, Z, ell, var)
    Lu = torch.linalg.cholesky(Ku + torch.eye(M, dtype=Ku.dtype) * JITTER)              # svpy.py:190
    A = torch.linalg.solve_triangular(Lu, rbf_K(Z, x, ell, var), upper=False)            # (Do,M,N)
    Us = softplus(p['Us']).T[:, :, None] if is_q_diag(p['Us'], Um) else tril_unpack(p['Us'], M)
    SK = Us @ Us.permute(0, 2, 1) - torch.eye(M, dtype=Ku.dtype).unsqueeze(0)
    B = torch.einsum('dme,den->dmn', SK, A)
    if full_cov:
        cov = rbf_K(x, x, ell, var) + torch.einsum('dme,dmn->den', A, B)
    else:
        cov = torch.diagonal(rbf_K(x, x, ell, var), dim1=1, dim2=2) + (A * B).sum(1)
    return torch.einsum('dmn,md->nd', A, Um), cov.permute(*reversed(range(cov.dim())))


def svgp_kl(Um, Us_packed):
    """svpy.py:144-175"""
    M = Um.shape[0]
    if is_q_diag(Us_packed, Um):                 # svpy.py:153-154,165-166
        Lq = Lq_diag = softplus(Us_packed)
        trace = torch.pow(Lq, 2).sum(dim=0, keepdim=True)
    else:
        Lq = torch.tril(tril_unpack(Us_packed, M))
        Lq_diag = torch.diagonal(Lq, dim1=1, dim2=2).t()
        trace = torch.pow(Lq, 2).sum(dim=(1, 2)).unsqueeze(0)
    mahalanobis = torch.pow(Um, 2).sum(dim=0, keepdim=True)
    logdet_qcov = torch.log(torch.pow(Lq_diag, 2)).sum(dim=0, keepdim=True)
    twoKL = 0.0 - logdet_qcov + mahalanobis + trace + (-torch.tensor(M))
    return 0.5 * twoKL.sum()


# ----------------------------------------------------------------------------
# ODE right-hand side and fixed-grid integration   model/core/flow.py
# ----------------------------------------------------------------------------
def ode_rhs(sv, c, order):
    """flow.py:27-45 (autonomous: t ignored)."""
    if order == 1:
        return gp_forward(sv, c)
    q = sv.shape[1] // 2
    return torch.cat([sv[:, q:], gp_forward(sv, c)], 1)


def odeint_fixed(f, y0, ts, method):
    """Restatement of torchdiffeq's fixed-grid solvers at the reference's call
    site flow.py:76-85 (grid = output times; atol/rtol ignored).  'euler':
    y1 = y0 + dt*f(y0).  'rk4' = 3/8 rule (torchdiffeq rk4_alt_step_func):
      k1=f(y0); k2=f(y0+dt*k1/3); k3=f(y0+dt*(k2-k1/3)); k4=f(y0+dt*(k1-k2+k3));
      y1 = y0 + (k1+3*(k2+k3)+k4)*dt*0.125.
    'midpoint': y1 = y0 + dt*f(y0 + dt/2*f(y0)).
    NOT pinned by reference code (torchdiffeq absent)."""
    ys = [y0]
    y = y0
    third = 1.0 / 3.0
    for j in range(len(ts) - 1):
        dt = ts[j + 1] - ts[j]
        k1 = f(y)
        if method == 'euler':
            dy = dt * k1
        elif method == 'rk4':
            k2 = f(y + dt * k1 * third)
            k3 = f(y + dt * (k2 - k1 * third))
            k4 = f(y + dt * (k1 - k2 + k3))
            dy = (k1 + 3 * (k2 + k3) + k4) * dt * 0.125
        elif method == 'midpoint':       # torchdiffeq Midpoint._step_func: dt * f(t0 + dt/2, y0 + f(t0, y0) * dt/2)
            dy = dt * f(y + k1 * (0.5 * dt))
        else:
            raise ValueError(method)
        y = y + dy
        ys.append(y)
    return torch.stack(ys, 0)


def flow_forward(z0, ts, c, order, method):
    """flow.py:68-86 -> (N,T,D) given a prebuilt cache."""
    zt = odeint_fixed(lambda y: ode_rhs(y, c, order), z0, ts, method)
    return zt.permute([1, 0, 2])


# ----------------------------------------------------------------------------
# conv VAE   model/core/vae.py
# ----------------------------------------------------------------------------
BN_EPS = 1e-5


def _bn_train(x, w, b):
    return F.batch_norm(x, None, None, w, b, training=True, momentum=0.1, eps=BN_EPS)


def encoder_forward(x, sd, prefix='vae.encoder.'):
    """vae.py:53-73 (BatchNorm in train mode, F11) -> mu, logvar (N,q)"""
    g = lambda k: sd[prefix + k]
    h = F.conv2d(x, g('cnn.0.weight'), g('cnn.0.bias'), stride=2, padding=2)
    h = F.relu(_bn_train(h, g('cnn.1.weight'), g('cnn.1.bias')))
    h = F.conv2d(h, g('cnn.3.weight'), g('cnn.3.bias'), stride=2, padding=2)
    h = F.relu(_bn_train(h, g('cnn.4.weight'), g('cnn.4.bias')))
    h = F.relu(F.conv2d(h, g('cnn.6.weight'), g('cnn.6.bias'), stride=2, padding=2))
    z = F.linear(h.flatten(1), g('fc.weight'), g('fc.bias'))
    return z.chunk(2, dim=-1)


def reparam(mu, logvar, eps):
    """vae.py:75-78"""
    return mu + torch.exp(0.5 * logvar) * eps


def decoder_forward(z, sd, prefix='vae.decoder.'):
    """vae.py:108-129 -> (prod(lead),1,28,28)"""
    g = lambda k: sd[prefix + k]
    s = F.linear(z.contiguous().view([int(np.prod(list(z.shape[:-1]))), z.shape[-1]]), g('fc.weight'), g('fc.bias'))
    h = s.view(s.size(0), s[0].numel() // 16, 4, 4)
    h = F.conv_transpose2d(h, g('decnn.1.weight'), g('decnn.1.bias'), stride=1, padding=0)
    h = F.relu(_bn_train(h, g('decnn.2.weight'), g('decnn.2.bias')))
    h = F.conv_transpose2d(h, g('decnn.4.weight'), g('decnn.4.bias'), stride=2, padding=1)
    h = F.relu(_bn_train(h, g('decnn.5.weight'), g('decnn.5.bias')))
    h = F.conv_transpose2d(h, g('decnn.7.weight'), g('decnn.7.bias'), stride=2, padding=1, output_padding=1)
    h = F.relu(_bn_train(h, g('decnn.8.weight'), g('decnn.8.bias')))
    h = F.conv_transpose2d(h, g('decnn.10.weight'), g('decnn.10.bias'), stride=1, padding=2)
    return torch.sigmoid(h)


def bernoulli_log_prob(X, Xrec, L):
    """vae.py:136-153 (no epsilon: torch.log never raises, F9)"""
    XL = X.repeat([L, 1, 1, 1, 1, 1])
    return torch.log(Xrec) * XL + torch.log(1 - Xrec) * (1 - XL)


def gauss_kl_std_normal(mu, logvar):
    """create_model.py:48-49 via torch.distributions.kl_divergence(Normal, Normal(0,1))"""
    from torch.distributions import Normal, kl_divergence
    q = Normal(mu, torch.exp(0.5 * logvar))
    p = Normal(torch.zeros(mu.shape[-1], dtype=mu.dtype), torch.ones(mu.shape[-1], dtype=mu.dtype))
    return kl_divergence(q, p).sum(-1)


# ----------------------------------------------------------------------------
# full model + loss   model/core/odegpvae.py, model/create_model.py
# ----------------------------------------------------------------------------
GP_KEYS = dict(raw_ell='flow.odefunc.diffeq.kern.unconstrained_lengthscales',
               raw_var='flow.odefunc.diffeq.kern.unconstrained_variance',
               Z='flow.odefunc.diffeq.inducing_loc.optvar',
               Um='flow.odefunc.diffeq.Um.optvar',
               Us='flow.odefunc.diffeq.Us_sqrt.optvar')


def gp_params_from_state_dict(sd):
    return {k: sd[v] for k, v in GP_KEYS.items()}


def model_forward(X, sd, noises, eps_s, eps_v, *, kernel, order, method, dt, v_steps=5, T_custom=None):
    """odegpvae.py:48-70.  noises: list (len L) of GP-draw noise dicts."""
    N, T = X.shape[0], X.shape[1]
    if T_custom:
        T = T_custom
    s_mu, s_logv = encoder_forward(X[:, 0], sd, 'vae.encoder.')
    z0 = reparam(s_mu, s_logv, eps_s)
    v_mu = v_logv = None
    if order == 2:
        v_mu, v_logv = encoder_forward(torch.squeeze(X[:, 0:v_steps]), sd, 'vae.encoder_v.')
        z0 = torch.concat([z0, reparam(v_mu, v_logv, eps_v)], dim=1)
    ts = dt * torch.arange(T, dtype=torch.float).to(z0.dtype)
    p = gp_params_from_state_dict(sd)
    ztL = torch.cat([flow_forward(z0, ts, build_cache(p, nz, kernel), order, method).unsqueeze(0)
                     for nz in noises], 0)
    L = len(noises)
    zdec = ztL if order == 1 else ztL[:, :, :, :ztL.shape[-1] // 2]
    Xrec = decoder_forward(zdec, sd).view([L, N, T, 1, 28, 28])
    return Xrec, ztL, (s_mu, s_logv), (v_mu, v_logv)


def compute_loss(X, sd, noises, eps_s, eps_v, *, kernel, order, method, dt, Ndata, v_steps=5):
    """create_model.py:37-73 -> dict(loss, nlhood, kl_reg, kl_u, Xrec, ztL, ...)"""
    Xrec, ztL, (s_mu, s_logv), (v_mu, v_logv) = model_forward(
        X, sd, noises, eps_s, eps_v, kernel=kernel, order=order, method=method, dt=dt, v_steps=v_steps)
    L = len(noises)
    mu = s_mu if v_mu is None else torch.cat((s_mu, v_mu), dim=1)
    logv = s_logv if v_logv is None else torch.cat((s_logv, v_logv), dim=1)
    kl_reg = gauss_kl_std_normal(mu, logv).mean()
    lhood = bernoulli_log_prob(X, Xrec, L).sum([2, 3, 4, 5]).mean(0).mean()
    p = gp_params_from_state_dict(sd)
    kl_u = svgp_kl(p['Um'], p['Us'])
    loss = -(lhood * Ndata - kl_reg * Ndata - kl_u)
    return dict(loss=loss, nlhood=-lhood, kl_reg=kl_reg, kl_u=kl_u, Xrec=Xrec, ztL=ztL,
                s_mu=s_mu, s_logv=s_logv, v_mu=v_mu, v_logv=v_logv)


def to_dtype(tree, dtype):
    """Cast every floating tensor of a (nested) dict/list to ``dtype`` (fp64 twin)."""
    if isinstance(tree, dict):
        return {k: to_dtype(v, dtype) for k, v in tree.items()}
    if isinstance(tree, (list, tuple)):
        return type(tree)(to_dtype(v, dtype) for v in tree)
    if torch.is_tensor(tree) and tree.is_floating_point():
        return tree.to(dtype)
    return tree
