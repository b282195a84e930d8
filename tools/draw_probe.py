"""Phase cycles of k_draw_lds (library built with -DDRAW_PROBE into tools/probe/) at the BASELINE RBF / tiny DF shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_gp_ode_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'probe', 'libgpode_hip_probe.so')
from vae_gp_ode_amd import ops
from oracle import gpode_oracle as O
g = torch.Generator().manual_seed(0)
for kernel, q, M in (('RBF', 6, 100), ('DF', 6, 16), ('RBF', 6, 180)):
    S = 256
    p = dict(raw_ell=O.invsoftplus(torch.full((q, q), 2.0)), raw_var=O.invsoftplus(torch.ones(q)), Z=torch.randn(M, q, generator=g),
             Um=0.1 * torch.randn(M, q, generator=g), Us=O.tril_pack(torch.stack([torch.eye(M)] * q) * 1e-3))
    nz = dict(eps_u=torch.randn(M, q, generator=g), rff_w=torch.randn(S if kernel == 'RBF' else 2 * S, q, generator=g),
              rff_eps=torch.randn(q, S, q, generator=g), rff_u=torch.rand(1, S, q, generator=g))
    args = [p[k].cuda() for k in ('raw_ell', 'raw_var', 'Z', 'Um', 'Us')] + [nz[k].cuda() for k in ('eps_u', 'rff_w', 'rff_eps', 'rff_u')]
    print(kernel, 'M =', M, flush=True)
    for _ in range(3):
        c = ops.cache_build(kernel, *args)
        torch.cuda.synchronize()
