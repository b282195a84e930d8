"""Model factory + ELBO (operator API of experiments/model/create_model.py)."""
import torch
from torch.distributions import kl_divergence as kl

from .core.flow import Flow
from .core.odegpvae import ODEGPVAE
from .core.svpy import SVGP_Layer
from .core.vae import VAE


def build_model(args):
    gp = SVGP_Layer(D_in=args.D_in, D_out=args.D_out, M=args.num_inducing, S=args.num_features,
                    dimwise=args.dimwise, q_diag=args.q_diag, device=args.device, kernel=args.kernel)
    flow = Flow(diffeq=gp, order=args.ode, solver=args.solver, use_adjoint=args.use_adjoint)
    vae = VAE(frames=args.frames, n_filt=args.n_filt, latent_dim=args.latent_dim, order=args.ode, device=args.device)
    return ODEGPVAE(flow=flow, vae=vae, num_observations=args.Ndata, order=args.ode, steps=args.frames, dt=args.dt)


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
    Xrec, (s0_mu, s0_logv), (v0_mu, v0_logv) = model(data, L)
    kl_rows = model.vae.encoder.kl_rows(s0_mu, s0_logv, v0_mu, v0_logv)               # (N,)
    lhood_rows = model.vae.decoder.log_prob_rowsum(data, Xrec, L)                     # (L, N)
    out = V.elbo_terms(lhood_rows, kl_rows, model.flow.kl(), model.num_observations)
    return out[0], out[1], out[2], out[3]


def compute_test_error(X, Xrec):
    assert list(X.shape) == list(Xrec.shape), f'incorrect shapes X: {list(X.shape)}, X_Rec: {list(Xrec.shape)}'
    return torch.mean((Xrec - X) ** 2)
