"""Conv VAE (operator API and state_dict layout of experiments/model/core/vae.py).

The ``nn.Sequential`` containers below exist for their parameters, buffers and ``state_dict`` keys
(``cnn.{0,3,6}``, ``fc``, ``decnn.{1,4,7,10}``, BatchNorm running statistics ...), which are the reference's,
so checkpoints interchange.  The arithmetic -- convolutions, transposed convolutions, training-mode
BatchNorm, ReLU/sigmoid, the linear layers and the Bernoulli log-likelihood, forward and backward -- runs in
the hand-written HIP kernels of csrc/vae_conv.hip through ``vae_ops``; nothing is dispatched to torch.nn.
BatchNorm runs on batch statistics exactly as the reference's training loop does (it never calls
``.eval()``); eval-mode BatchNorm (only reached via ``--pretrained`` / ``VAE.test``) is not built.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.distributions import Normal

from ..misc.torch_utils import UnFlatten
from ... import vae_ops as V


def _bn(x, bn, relu):
    if not bn.training:     # module.eval(): the frozen pre-trained VAE of main.py:157-163, VAE.test
        return V.batch_norm_eval(x, bn, relu)
    return V.batch_norm_train(x, bn, relu)

EPSILON = 1e-3


class Encoder(nn.Module):
    def __init__(self, latent_dim=16, n_filt=8, frames=1):
        super().__init__()
        self.cnn = nn.Sequential(
            nn.Conv2d(frames, n_filt, kernel_size=5, stride=2, padding=(2, 2)),          # 28 -> 14
            nn.BatchNorm2d(n_filt), nn.ReLU(),
            nn.Conv2d(n_filt, n_filt * 2, kernel_size=5, stride=2, padding=(2, 2)),     # 14 -> 7
            nn.BatchNorm2d(n_filt * 2), nn.ReLU(),
            nn.Conv2d(n_filt * 2, n_filt * 4, kernel_size=5, stride=2, padding=(2, 2)),  # 7 -> 4
            nn.ReLU(), nn.Flatten())
        self.fc = nn.Linear(n_filt * 4 ** 3, 2 * latent_dim)

    def forward(self, x):
        c = self.cnn
        h = V.conv2d(x, c[0].weight, c[0].bias, 2, 2)        # (a batch-strided slice of the minibatch is read in place)
        h = _bn(h, c[1], relu=True)
        h = V.conv2d(h, c[3].weight, c[3].bias, 2, 2)
        h = _bn(h, c[4], relu=True)
        h = V.conv2d(h, c[6].weight, c[6].bias, 2, 2)
        z = V.linear_relu_in(h.flatten(1), self.fc.weight, self.fc.bias)   # nn.ReLU + nn.Flatten + fc: the ReLU rides in the layer's loads
        return z.chunk(2, dim=-1)

    @staticmethod
    def forward_pair(enc_a, xa, enc_b, xb):
        """enc_a(xa), enc_b(xb) layer by layer in lockstep, so that the same-depth BatchNorm layers of the two encoders exchange
        their cross-rank statistics in ONE packed all-gather per direction (data parallelism with global-minibatch BatchNorm only;
        vae_ops.set_pack_bn_gathers).  Same kernels on the same values as two separate forward() calls."""
        ca, cb = enc_a.cnn, enc_b.cnn
        ha = V.conv2d(xa.contiguous(), ca[0].weight, ca[0].bias, 2, 2)
        hb = V.conv2d(xb.contiguous(), cb[0].weight, cb[0].bias, 2, 2)
        ha, hb = V.batch_norm_train_pair(ha, ca[1], hb, cb[1], True)
        ha = V.conv2d(ha, ca[3].weight, ca[3].bias, 2, 2)
        hb = V.conv2d(hb, cb[3].weight, cb[3].bias, 2, 2)
        ha, hb = V.batch_norm_train_pair(ha, ca[4], hb, cb[4], True)
        ha = V.relu(V.conv2d(ha, ca[6].weight, ca[6].bias, 2, 2))
        hb = V.relu(V.conv2d(hb, cb[6].weight, cb[6].bias, 2, 2))
        za = V.linear(ha.flatten(1), enc_a.fc.weight, enc_a.fc.bias)
        zb = V.linear(hb.flatten(1), enc_b.fc.weight, enc_b.fc.bias)
        return za.chunk(2, dim=-1), zb.chunk(2, dim=-1)

    def sample(self, mu, logvar):
        """Reparameterised draw (vae.py:75-78).  ``next_eps`` (if set) replaces the N(0,1) draw once --
        used for parity tests and for data-parallel runs that must share the draw."""
        eps = getattr(self, 'next_eps', None)
        if eps is None:
            src = getattr(self, 'eps_source', None)  # noise.install_device_noise: the library's generator instead of torch's
            eps = src.normal(tuple(mu.shape), mu.device) if src is not None and mu.is_cuda else torch.randn_like(mu)
        self.next_eps = None
        return V.reparam(mu, logvar, eps.to(mu))     # mu + exp(logvar / 2) * eps, one launch

    def kl_rows(self, mu_s, logvar_s, mu_v=None, logvar_v=None):
        """kl_divergence(q_dist(...), N(0, I)).sum(-1) (create_model.py:47-49) without building the distributions: (N,)."""
        kl = V.normal_kl_rows(mu_s, logvar_s)         # the KL of a factorised Gaussian is additive over (s, v)
        return kl if mu_v is None else kl + V.normal_kl_rows(mu_v, logvar_v)

    def q_dist(self, mu_s, logvar_s, mu_v=None, logvar_v=None):
        if mu_v is not None:
            mu_s, logvar_s = torch.cat((mu_s, mu_v), dim=1), torch.cat((logvar_s, logvar_v), dim=1)
        # validate_args=False: the default argument check reads a device boolean back to the host (a stream
        # synchronisation per step, and not capturable into a HIP graph); exp() cannot produce an invalid scale
        return Normal(mu_s, torch.exp(0.5 * logvar_s), validate_args=False)

    @property
    def device(self):
        return next(self.parameters()).device


class Decoder(nn.Module):
    def __init__(self, latent_dim=16, n_filt=8, distribution='bernoulli'):
        super().__init__()
        self.distribution = distribution
        h_dim = n_filt * 4 ** 3
        self.fc = nn.Linear(latent_dim, h_dim)
        self.decnn = nn.Sequential(
            UnFlatten(4),
            nn.ConvTranspose2d(h_dim // 16, n_filt * 8, kernel_size=3, stride=1, padding=(0, 0)),  # 4 -> 6
            nn.BatchNorm2d(n_filt * 8), nn.ReLU(),
            nn.ConvTranspose2d(n_filt * 8, n_filt * 4, kernel_size=5, stride=2, padding=(1, 1)),   # 6 -> 13
            nn.BatchNorm2d(n_filt * 4), nn.ReLU(),
            nn.ConvTranspose2d(n_filt * 4, n_filt * 2, kernel_size=5, stride=2, padding=(1, 1), output_padding=(1, 1)),  # 13 -> 28
            nn.BatchNorm2d(n_filt * 2), nn.ReLU(),
            nn.ConvTranspose2d(n_filt * 2, 1, kernel_size=5, stride=1, padding=(2, 2)),
            nn.Sigmoid())

    def forward(self, x, logits=False):
        """``logits=True`` stops in front of the final nn.Sigmoid (create_model.compute_loss applies it fused with the likelihood)."""
        out = (lambda a: a) if logits else V.sigmoid
        flat = x.contiguous().view([int(np.prod(list(x.shape[:-1]))), x.shape[-1]])
        d = self.decnn
        h = V.linear(flat, self.fc.weight, self.fc.bias)
        h = h.view(h.size(0), h[0].numel() // 16, 4, 4)                                   # UnFlatten(4)
        if self.training and all(d[i].training for i in (2, 5, 8)):
            # training mode: each BatchNorm + ReLU is folded into the input staging of the transposed convolution that
            # consumes it -- the normalised activations (332 MB per step at the benchmark size) never go through HBM
            # (stats_for: the statistics of each output are summed by the convolution that stores it, for the BatchNorm behind it)
            c = V.conv_transpose2d(h, d[1].weight, d[1].bias, 1, 0, stats_for=d[2])                   # 4 -> 6
            c = V.bn_relu_conv_transpose2d(c, d[2], d[4].weight, d[4].bias, 2, 1, stats_for=d[5])     # 6 -> 13
            c = V.bn_relu_conv_transpose2d(c, d[5], d[7].weight, d[7].bias, 2, 1, 1, stats_for=d[8])  # 13 -> 28
            return out(V.bn_relu_conv_transpose2d(c, d[8], d[10].weight, d[10].bias, 1, 2))
        h = _bn(V.conv_transpose2d(h, d[1].weight, d[1].bias, 1, 0), d[2], relu=True)     # 4 -> 6
        h = _bn(V.conv_transpose2d(h, d[4].weight, d[4].bias, 2, 1), d[5], relu=True)     # 6 -> 13
        h = _bn(V.conv_transpose2d(h, d[7].weight, d[7].bias, 2, 1, 1), d[8], relu=True)  # 13 -> 28
        return out(V.conv_transpose2d(h, d[10].weight, d[10].bias, 1, 2))

    @property
    def device(self):
        return next(self.parameters()).device

    def log_prob(self, x, z, L=1, pretrain=False):
        """Bernoulli log-likelihood of targets x under reconstructions z (vae.py:136-153); no epsilon."""
        if self.distribution != 'bernoulli':
            raise ValueError('Currently only bernoulli dist implemented')
        return V.bernoulli_loglik(x, z)  # x is broadcast over the L leading copies of z inside the kernel

    def log_prob_rowsum(self, x, z, L=1):
        """sum over (T,c,h,w) of log_prob, shape (L,N): the reduction create_model.elbo applies, fused."""
        N = x.shape[0]
        return V.bernoulli_loglik_rowsum(x, z, L * N).view(L, N)


class VAE(nn.Module):
    def __init__(self, frames=1, n_filt=8, latent_dim=8, device='cpu', order=1, distribution='bernoulli'):
        super().__init__()
        self.encoder = Encoder(latent_dim, n_filt).to(device)
        self.decoder = Decoder(latent_dim, n_filt, distribution).to(device)
        self.prior = Normal(torch.zeros(latent_dim).to(device), torch.ones(latent_dim).to(device), validate_args=False)
        if order == 2:
            self.encoder_v = Encoder(latent_dim, n_filt, frames).to(device)
            self.prior = Normal(torch.zeros(latent_dim * 2).to(device), torch.ones(latent_dim * 2).to(device), validate_args=False)
        self.latent_dim = latent_dim
        self.order = order

    def print_summary(self):
        print(self.encoder)
        print(self.decoder)

    def save(self, encoder_path=None, decoder_path=None):
        torch.save(self.encoder.state_dict(), encoder_path)
        torch.save(self.decoder.state_dict(), decoder_path)

    def test(self, x):
        self.encoder.eval()
        self.decoder.eval()
        mu, logv = self.encoder(x)
        return self.decoder(self.encoder.sample(mu, logv))
