"""Conv VAE (operator API and state_dict layout of experiments/model/core/vae.py).

ROUND-1 STATUS: the encoder/decoder modules below keep the reference's parameter names and shapes
(``cnn.{0,3,6}``, ``fc``, ``decnn.{1,4,7,10}`` ...) so checkpoints interchange, but their arithmetic is
still dispatched by torch.nn (MIOpen/rocBLAS on the GPU) -- the hand-written implicit-GEMM conv + BN +
Bernoulli log-likelihood kernels are the next row of the build plan ("what comes next").
"""
import numpy as np
import torch
import torch.nn as nn
from torch.distributions import Normal

from ..misc.torch_utils import UnFlatten

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
        return self.fc(self.cnn(x)).chunk(2, dim=-1)

    def sample(self, mu, logvar):
        std = torch.exp(0.5 * logvar)
        return mu + std * torch.randn_like(std)

    def q_dist(self, mu_s, logvar_s, mu_v=None, logvar_v=None):
        if mu_v is not None:
            mu_s, logvar_s = torch.cat((mu_s, mu_v), dim=1), torch.cat((logvar_s, logvar_v), dim=1)
        return Normal(mu_s, torch.exp(0.5 * logvar_s))

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

    def forward(self, x):
        flat = x.contiguous().view([int(np.prod(list(x.shape[:-1]))), x.shape[-1]])
        return self.decnn(self.fc(flat))

    @property
    def device(self):
        return next(self.parameters()).device

    def log_prob(self, x, z, L=1, pretrain=False):
        """Bernoulli log-likelihood of targets x under reconstructions z (vae.py:136-153); no epsilon."""
        XL = x if pretrain else x.repeat([L, 1, 1, 1, 1, 1])
        if self.distribution != 'bernoulli':
            raise ValueError('Currently only bernoulli dist implemented')
        return torch.log(z) * XL + torch.log(1 - z) * (1 - XL)


class VAE(nn.Module):
    def __init__(self, frames=1, n_filt=8, latent_dim=8, device='cpu', order=1, distribution='bernoulli'):
        super().__init__()
        self.encoder = Encoder(latent_dim, n_filt).to(device)
        self.decoder = Decoder(latent_dim, n_filt, distribution).to(device)
        self.prior = Normal(torch.zeros(latent_dim).to(device), torch.ones(latent_dim).to(device))
        if order == 2:
            self.encoder_v = Encoder(latent_dim, n_filt, frames).to(device)
            self.prior = Normal(torch.zeros(latent_dim * 2).to(device), torch.ones(latent_dim * 2).to(device))
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
