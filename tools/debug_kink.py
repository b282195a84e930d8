# is there a ReLU pre-activation within fp32 round-off of zero in this fixture? (HIP pipeline vs fp64 pipeline, each end to end)
import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden, sub
from test_gpu_model import make_model
from vae_gp_ode_amd import vae_ops as V
name, kw, L = 'model_rbf1_tiny', {}, 2
m, g = make_model(name, kw, L)
acts = []
orig_bn, orig_relu = V.batch_norm_train, V.relu
def bn(x, mod, relu):
    y = orig_bn(x, mod, relu); acts.append(y.detach().double().cpu()); return y
def rl(x):
    y = orig_relu(x); acts.append(y.detach().double().cpu()); return y
V.batch_norm_train, V.relu = bn, rl
with torch.no_grad():
    m(g['X'].cuda(), L)
# fp64 pipeline with the oracle
from oracle import gpode_oracle as O
sd = {k: (v.double() if v.is_floating_point() else v) for k, v in sub(g, 'sd.').items()}
ref = []
_old = F.relu
def spy_relu(x, *a, **k):
    y = _old(x, *a, **k); ref.append(y.detach()); return y
O.F.relu = spy_relu
with torch.no_grad():
    O.model_forward(g['X'].double(), sd, [O.to_dtype(sub(g, 'noise%d.' % l), torch.float64) for l in range(L)], g['eps_s'].double(), None,
                    kernel='RBF', order=1, method='rk4', dt=0.1)
O.F.relu = _old
print(len(acts), len(ref))
for i, (a, b) in enumerate(zip(acts, ref)):
    flips = (a > 0) != (b > 0)
    print('relu %d shape %s flips %d' % (i, tuple(a.shape), int(flips.sum())), 'values at flips:', a[flips].tolist()[:4], b[flips].tolist()[:4])
