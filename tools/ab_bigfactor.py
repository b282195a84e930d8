import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import conftest
from oracle import gpode_oracle as O
import test_gpu_backward as T
from test_gpu_forward import relerr
from vae_gp_ode_amd.model.core.flow import Flow
from vae_gp_ode_amd.model.core.svpy import SVGP_Layer
D_ = int(os.environ.get('DD', '3')); kernel,Di,Do,order,M,S,method = 'RBF',D_,D_,1,int(os.environ.get('MM','1056')),64,'euler'
N,T_=5,4
p, nz, z0, ts, gw = T.synthetic_gp(kernel, Di, Do, M, S, N, T_, seed=1000 + M + S + Di)
gp = SVGP_Layer(Di, Do, M, S, kernel=kernel).cuda()
with torch.no_grad():
    gp.kern.unconstrained_lengthscales.copy_(p['raw_ell']); gp.kern.unconstrained_variance.copy_(p['raw_var'])
    gp.inducing_loc.optvar.copy_(p['Z']); gp.Um.optvar.copy_(p['Um']); gp.Us_sqrt.optvar.copy_(p['Us'])
flow = Flow(gp, order=order, solver=method).cuda()
gp.set_noise({k: v.cuda() for k, v in nz.items()})
zg = z0.cuda().requires_grad_(True)
zt = flow(zg, ts.cuda()); (zt * gw.cuda()).sum().backward()
got = {'raw_ell': gp.kern.unconstrained_lengthscales.grad, 'raw_var': gp.kern.unconstrained_variance.grad,
       'Z': gp.inducing_loc.optvar.grad, 'Um': gp.Um.optvar.grad, 'Us': gp.Us_sqrt.optvar.grad, 'z0': zg.grad}
def oracle(dtype):
    q = {k: v.to(dtype).clone().requires_grad_(True) for k, v in p.items()}
    c = O.build_cache(q, O.to_dtype(nz, dtype), kernel)
    z = z0.to(dtype).clone().requires_grad_(True)
    out = O.flow_forward(z, ts.to(dtype), c, order, method)
    (out * gw.to(dtype)).sum().backward()
    return out.detach(), dict({k: v.grad for k, v in q.items()}, z0=z.grad)
z64, g64 = oracle(torch.float64); z32, g32 = oracle(torch.float32)
print('M', M, 'small-kernels' if os.environ.get('GPODE_SMALL_FACTOR_KERNELS') == '1' else 'big-kernels', 'zt %.1e (cpu32 %.1e)' % (relerr(zt, z64), relerr(z32, z64)))
for k in got:
    print('  %-8s hip %.2e   cpu32 %.2e' % (k, relerr(got[k], g64[k]), relerr(g32[k], g64[k])))
