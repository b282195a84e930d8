"""A few launches of the fused decnn.10 / BatchNorm backward passes at 4096 images, for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_gp_ode_amd import vae_ops as V
B = 4096
g = torch.Generator().manual_seed(0)
c = (torch.randn(B, 16, 28, 28, generator=g) * 1.3 + 0.2).cuda()
gam, bet = (torch.rand(16, generator=g) + 0.5).cuda(), (torch.randn(16, generator=g) * 0.3).cuda()
w, gy = (torch.randn(16, 1, 5, 5, generator=g) * 0.05).cuda(), torch.randn(B, 1, 28, 28, generator=g).cuda()
mean = c.mean((0, 2, 3))
invstd = torch.rsqrt(c.var((0, 2, 3), unbiased=False) + 1e-5)
for _ in range(3):
    out = V._dec10_bn_bwd(None, c, gy, w, gam, bet, mean, invstd)
torch.cuda.synchronize()
print('ok', float(out[0].abs().sum()))
