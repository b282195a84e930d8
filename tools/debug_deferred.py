"""Which gradients differ between immediate and deferred final reductions of one backward pass?"""
import types
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_gp_ode_amd import ops, vae_ops
from vae_gp_ode_amd.model.core.initialization import initialize_and_fix_kernel_parameters
from vae_gp_ode_amd.model.core.noise import DeviceNoise
from vae_gp_ode_amd.model.create_model import build_model, compute_loss
from vae_gp_ode_amd.model.misc.torch_utils import seed_everything

seed_everything(4)
args = types.SimpleNamespace(D_in=6, D_out=6, num_inducing=32, num_features=64, dimwise=True, q_diag=False, device='cuda',
                             kernel='DF', ode=1, solver='rk4', use_adjoint=False, frames=5, n_filt=8, latent_dim=6, Ndata=64, dt=0.1)
m = build_model(args).cuda()
initialize_and_fix_kernel_parameters(m, 2.0, 1.0)
X = torch.rand(16, 8, 1, 28, 28, device='cuda')
fixed = DeviceNoise(9).draw('DF', 6, 6, 32, 64, 'cuda')


class FixedNoise:
    def draw(self, *a):
        return fixed


m.flow.odefunc.diffeq.noise_source = FixedNoise()
eps = torch.randn(16, 6, device='cuda')
names = [n for n, _ in m.named_parameters()]


def grads(deferred):
    vae_ops.set_deferred_reductions(deferred)
    for p in m.parameters():
        p.grad = None
    m.vae.encoder.next_eps = eps
    loss, *_ = compute_loss(m, X, 1)
    loss.backward()
    torch.cuda.synchronize()
    return [p.grad.clone() for p in m.parameters()]


bn0 = [b.clone() for b in m.buffers()]
a = grads(False)
for b, v in zip(m.buffers(), bn0):
    b.copy_(v)
b = grads(True)
for b_, v in zip(m.buffers(), bn0):
    b_.copy_(v)
c = grads(True)
for n, x, y, z in zip(names, a, b, c):
    d1, d2 = float((x - y).abs().max()), float((x - z).abs().max())
    print('%-45s %-18s immediate-vs-deferred %.3e  (second deferred pass %.3e)  |g| %.3e' % (n, tuple(x.shape), d1, d2, float(x.abs().max())))
