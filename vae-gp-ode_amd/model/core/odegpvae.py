"""Composite model: encode -> L x integrate -> decode (operator API of experiments/model/core/odegpvae.py)."""
import torch
import torch.nn as nn


class ODEGPVAE(nn.Module):
    def __init__(self, flow, vae, num_observations, steps, order=2, dt=0.1):
        super().__init__()
        self.flow = flow
        self.vae = vae
        self.num_observations = num_observations
        self.dt = dt
        self.v_steps = steps
        self.order = order

    def build_decoding(self, ztL, dims):
        """ztL (L,N,T,order*q) -> Xrec (L,N,T,nc,d,d); only positions are decoded for order 2 (odegpvae.py:18-35)."""
        L, N, T, nc, d, _ = dims
        lat = ztL if self.order == 1 else ztL[..., :ztL.shape[-1] // 2]
        return self.vae.decoder(lat).view([L, N, T, nc, d, d])

    def sample_trajectories(self, z0, T, L=1):
        """L independent function draws, each shared by the whole minibatch (odegpvae.py:37-45)."""
        key = (T, str(z0.device))
        if getattr(self, '_ts_key', None) != key:            # the grid dt * arange(T) is a constant of the run: build it once
            self._ts, self._ts_key = self.dt * torch.arange(T, dtype=torch.float, device=z0.device), key
        ts = self._ts
        if L == 1:
            return self.flow(z0, ts).unsqueeze(0)
        return torch.stack([self.flow(z0, ts) for _ in range(L)], 0)

    def forward(self, X, L=1, T_custom=None):
        N, T, nc, d, _ = X.shape
        if T_custom:
            T = T_custom
        enc = self.vae.encoder
        gp = self.flow.odefunc.diffeq
        if L == 1 and hasattr(gp, 'prebuild_cache'):
            gp.prebuild_cache()                      # overlap mode only: the draw's cache builds next to the encoder
        s0_mu, s0_logv = enc(X[:, 0])
        z0 = enc.sample(mu=s0_mu, logvar=s0_logv)
        v0_mu = v0_logv = None
        if self.order == 2:
            enc_v = self.vae.encoder_v
            v0_mu, v0_logv = enc_v(torch.squeeze(X[:, 0:self.v_steps]))
            z0 = torch.concat([z0, enc_v.sample(mu=v0_mu, logvar=v0_logv)], dim=1)
        ztL = self.sample_trajectories(z0, T, L)
        Xrec = self.build_decoding(ztL, (L, N, T, nc, d, d))
        return Xrec, (s0_mu, s0_logv), (v0_mu, v0_logv)
