"""configs[4] forward at N=128, T=64: hip vs fp64 oracle vs fp32 oracle, error per time index."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from oracle import gpode_oracle as O
from test_gpu_baseline_sizes import _model_and_draw
torch.set_num_threads(16)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = dict(kernel='DF', ode=1, q=16, M=512, S=256, T=T, N=128)
m, X, nz, eps_s, _ = _model_and_draw(cfg)
sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
with torch.no_grad():
    z0h, _, _ = (lambda: (m.vae.encoder.__setattr__('next_eps', eps_s.cuda()), m.encode_initial_state(X.cuda()))[1])()
    gp = m.flow.odefunc.diffeq
    gp.set_noise({k: v.cuda() for k, v in nz.items()})
    zt = m.sample_trajectories(z0h, T, 1)[0].double().cpu()
    res = {}
    for dt in (torch.float64, torch.float32):
        s = O.to_dtype(sd, dt)
        mu, lv = O.encoder_forward(X[:, 0].to(dt), s, 'vae.encoder.')
        z0 = O.reparam(mu, lv, eps_s.to(dt))
        p = O.gp_params_from_state_dict(s)
        c = O.build_cache(p, O.to_dtype(nz, dt), 'DF')
        res[dt] = (O.flow_forward(z0, (0.1 * torch.arange(T, dtype=torch.float)).to(dt), c, 1, 'rk4').double(), c)
    z64, z32 = res[torch.float64][0], res[torch.float32][0]
    print('z0: hip vs 64 %.1e' % ((z0h.double().cpu() - z64[:, 0]).abs().max() / z64[:, 0].abs().max()))
    print('nu: hip vs 64 %.1e ; fp32 oracle vs 64 %.1e' % (((gp.cache.nu.double().cpu().flatten() - res[torch.float64][1]['nu'].flatten()).abs().max() / res[torch.float64][1]['nu'].abs().max()).item(),
          ((res[torch.float32][1]['nu'].double().flatten() - res[torch.float64][1]['nu'].flatten()).abs().max() / res[torch.float64][1]['nu'].abs().max()).item()))
    for t in list(range(0, T, max(1, T // 16))) + [T - 1]:
        sc = z64[:, t].abs().max()
        print('t=%2d |z|max %.2f  hip-64 %.1e   32-64 %.1e' % (t, sc, (zt[:, t] - z64[:, t]).abs().max() / sc, (z32[:, t] - z64[:, t]).abs().max() / sc))
