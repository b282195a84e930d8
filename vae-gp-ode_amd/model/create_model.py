"""Model factory + ELBO (operator API of experiments/model/create_model.py)."""
import os

import torch
from torch.distributions import kl_divergence as kl

from .core.flow import Flow
from .core.odegpvae import ODEGPVAE
from .core.svpy import SVGP_Layer
from .core.vae import VAE


_FUSED_LOSS = os.environ.get('GPODE_UNFUSED_LOSS', '0') != '1'


def build_model(args):
    """GP vector field -> Flow -> VAE -> ODEGPVAE from the argument namespace of main.py (create_model.py:16-33 reads the
    same fields).  The construction order fixes the order of the numpy draws that initialise the GP parameters."""
    a = args
    field = SVGP_Layer(a.D_in, a.D_out, a.num_inducing, a.num_features, q_diag=a.q_diag, dimwise=a.dimwise, device=a.device,
                       kernel=a.kernel)
    return ODEGPVAE(flow=Flow(field, order=a.ode, solver=a.solver, use_adjoint=a.use_adjoint),
                    vae=VAE(frames=a.frames, n_filt=a.n_filt, latent_dim=a.latent_dim, device=a.device, order=a.ode),
                    num_observations=a.Ndata, steps=a.frames, order=a.ode, dt=a.dt)


def elbo(model, X, Xrec, s0_mu, s0_logv, v0_mu, v0_logv, L):
    """-> (mean log-likelihood per sequence, mean KL(q(z0)||p), KL(q(u)||p)) (create_model.py:37-58)."""
    q = model.vae.encoder.q_dist(s0_mu, s0_logv, v0_mu, v0_logv)
    kl_reg = kl(q, model.vae.prior).sum(-1)
    lhood = model.vae.decoder.log_prob_rowsum(X, Xrec, L).mean(0)  # == log_prob(...).sum([2,3,4,5]).mean(0)
    return lhood.mean(), kl_reg.mean(), model.flow.kl()


def compute_loss(model, data, L):
    """-> (loss, nll, kl_reg, kl_u) (create_model.py:61-73).  Same terms as elbo() above; the per-row pieces go through
    three fused launches (KL rows, likelihood row sums, loss algebra) instead of ~25 elementwise ones."""
    from .. import vae_ops as V
    field = model.flow.odefunc.diffeq
    if _FUSED_LOSS and hasattr(field, 'us_packed') and model.vae.decoder.distribution == 'bernoulli':
        # the decoder stops at its logits; sigmoid + likelihood row sums are one pass over them, and the three KL / mean / loss
        # launches one more (gpode_sigmoid_loglik_fwd, gpode_elbo_all_fwd)
        logits, (s0_mu, s0_logv), (v0_mu, v0_logv) = model(data, L, logits=True)
        lpart, _ = V.sigmoid_loglik_parts(data, logits, L * data.shape[0])
        return V.elbo_all(lpart, s0_mu, s0_logv, v0_mu, v0_logv, field.Um.optvar, field.us_packed(), field.M, model.num_observations)
    Xrec, (s0_mu, s0_logv), (v0_mu, v0_logv) = model(data, L)
    kl_rows = model.vae.encoder.kl_rows(s0_mu, s0_logv, v0_mu, v0_logv)               # (N,)
    lhood_rows = model.vae.decoder.log_prob_rowsum(data, Xrec, L)                     # (L, N)
    out = V.elbo_terms(lhood_rows, kl_rows, model.flow.kl(), model.num_observations)
    return out[0], out[1], out[2], out[3]


_ONE = {}


def backward(loss):
    """``loss.backward()`` (main.py:210) with the root gradient handed over: autograd otherwise fills a fresh ones-tensor in every
    step -- one more few-microsecond launch on the chain of a step that is bound by the number of such launches."""
    key = (loss.device, loss.dtype)
    if key not in _ONE:
        _ONE[key] = torch.ones((), dtype=loss.dtype, device=loss.device)
    loss.backward(gradient=_ONE[key])


def compute_test_error(X, Xrec):
    assert list(X.shape) == list(Xrec.shape), f'incorrect shapes X: {list(X.shape)}, X_Rec: {list(Xrec.shape)}'
    return torch.mean((Xrec - X) ** 2)
