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

    def build_decoding(self, ztL, dims, logits=False):
        """ztL (L,N,T,order*q) -> Xrec (L,N,T,nc,d,d); only positions are decoded for order 2 (odegpvae.py:18-35)."""
        L, N, T, nc, d, _ = dims
        lat = ztL if self.order == 1 else ztL[..., :ztL.shape[-1] // 2]
        return (self.vae.decoder(lat, logits=True) if logits else self.vae.decoder(lat)).view([L, N, T, nc, d, d])

    def sample_trajectories(self, z0, T, L=1):
        """L independent function draws, each shared by the whole minibatch (odegpvae.py:37-45)."""
        key = (T, str(z0.device))
        if getattr(self, '_ts_key', None) != key:            # the grid dt * arange(T) is a constant of the run: build it once
            self._ts, self._ts_key = self.dt * torch.arange(T, dtype=torch.float, device=z0.device), key
        ts = self._ts
        if L == 1:
            return self.flow(z0, ts).unsqueeze(0)
        field = self.flow.odefunc.diffeq
        if getattr(field, 'batched_draws_supported', lambda: False)():
            # the L draws in ONE pass: K_uu factored once, one rollout launch over L * N trajectories, one reverse sweep
            return self.flow(z0, ts, draws=L)
        return torch.stack([self.flow(z0, ts) for _ in range(L)], 0)

    def _pair_encoders(self):
        from ... import vae_ops
        enc, vel = self.vae.encoder, self.vae.encoder_v
        return (vae_ops.pack_bn_gathers() and enc.training and vel.training and
                all(m.training for m in (enc.cnn[1], enc.cnn[4], vel.cnn[1], vel.cnn[4])))

    def encode_initial_state(self, X):
        """q(z0 | X): position code from the first frame, velocity code (order 2) from the first ``v_steps`` frames stacked as
        channels; returns the reparameterised sample and the (mean, log-variance) pairs the ELBO needs (odegpvae.py:55-63)."""
        pos = self.vae.encoder
        if self.order == 2 and self._pair_encoders():
            # data parallelism with global-minibatch BatchNorm: the two encoders run in lockstep and share their statistics exchanges
            vel = self.vae.encoder_v
            (mu_s, logv_s), (mu_v, logv_v) = pos.forward_pair(pos, X[:, 0], vel, torch.squeeze(X[:, 0:self.v_steps]))
            z0 = pos.sample(mu=mu_s, logvar=logv_s)
            return torch.concat([z0, vel.sample(mu=mu_v, logvar=logv_v)], dim=1), (mu_s, logv_s), (mu_v, logv_v)
        mu_s, logv_s = pos(X[:, 0])
        z0 = pos.sample(mu=mu_s, logvar=logv_s)
        if self.order == 1:
            return z0, (mu_s, logv_s), (None, None)
        vel = self.vae.encoder_v
        mu_v, logv_v = vel(torch.squeeze(X[:, 0:self.v_steps]))
        return torch.concat([z0, vel.sample(mu=mu_v, logvar=logv_v)], dim=1), (mu_s, logv_s), (mu_v, logv_v)

    def forward(self, X, L=1, T_custom=None, logits=False):
        """X (N,T,nc,d,d) -> (Xrec (L,N,T',nc,d,d), (mu_s, logv_s), (mu_v, logv_v)); T' = T_custom or T (odegpvae.py:48-70)."""
        N, T, nc, d, _ = X.shape
        horizon = T_custom if T_custom else T
        field = self.flow.odefunc.diffeq
        src = getattr(self.vae.encoder, 'eps_source', None)
        if src is not None and src is getattr(field, 'noise_source', None) and hasattr(field, 'predraw'):
            # device noise: the function draw(s) AND the encoders' reparameterisation draws in one launch, ahead of the encoder
            # (the same order of draws whether or not the cache build then runs on the side stream)
            src.reserve(N * (self.vae.encoder.fc.out_features // 2) * self.order)
            field.predraw(L if L > 1 and field.batched_draws_supported() else None)
        if hasattr(field, 'prebuild_cache') and (L == 1 or field.batched_draws_supported()):
            field.prebuild_cache(None if L == 1 else L)   # overlap mode only: the cache of the draw(s) builds next to the encoder
        z0, code_s, code_v = self.encode_initial_state(X)
        ztL = self.sample_trajectories(z0, horizon, L)
        return self.build_decoding(ztL, (L, N, horizon, nc, d, d), logits), code_s, code_v
