"""Locate the wrong elements of the decnn.10 fused BatchNorm stage backward at 8192 images (tests/test_gpu_baseline_sizes.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from vae_gp_ode_amd import vae_ops as V

torch.set_num_threads(16)
B, C, H = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 16, 28
g = torch.Generator().manual_seed(B + 7 * C)
c = torch.randn(B, C, H, H, generator=g) * 1.3 + 0.2
gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
w, b = torch.randn(16, 1, 5, 5, generator=g) * 0.05, torch.randn(1, generator=g) * 0.1
gy = torch.randn(B, 1, H, H, generator=g)

def rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max()).item()

# 1. plain BatchNorm backward at this shape vs torch fp64
x = c.cuda().requires_grad_(True)
gm, bt = gam.cuda().requires_grad_(True), bet.cuda().requires_grad_(True)
y = V._BatchNormTrain.apply(x, gm, bt, None, None, 0.1, 1e-5, 1)
gyy = torch.randn(B, C, H, H, generator=g)
y.backward(gyy.cuda())
x64 = c.double().requires_grad_(True); g64 = gam.double().requires_grad_(True); b64 = bet.double().requires_grad_(True)
y64 = F.relu(F.batch_norm(x64, None, None, g64, b64, True, 0.1, 1e-5))
y64.backward(gyy.double())
print('plain BN: y %.1e gx %.1e ggamma %.1e gbeta %.1e' % (rel(y, y64), rel(x.grad, x64.grad), rel(gm.grad, g64.grad), rel(bt.grad, b64.grad)))
per_img = (x.grad.double().cpu() - x64.grad).abs().amax(dim=(1, 2, 3))
bad = (per_img > 1e-4 * x64.grad.abs().max()).nonzero().flatten()
print('plain BN: images with wrong gx:', bad.numel(), bad[:40].tolist())

# 2. the fused stage, twice
outs = []
for rep in range(2):
    bn = torch.nn.BatchNorm2d(C).cuda()
    with torch.no_grad():
        bn.weight.copy_(gam); bn.bias.copy_(bet)
    a = [t.cuda().requires_grad_(True) for t in (c, w, b)]
    yy = V.bn_relu_conv_transpose2d(a[0], bn, a[1], a[2], 1, 2, 0)
    yy.backward(gy.cuda())
    outs.append((a[0].grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()))
print('fused stage run-to-run: gc equal', torch.equal(outs[0][0], outs[1][0]))
ref = torch.nn.BatchNorm2d(C).double()
with torch.no_grad():
    ref.weight.copy_(gam); ref.bias.copy_(bet)
a64 = [t.double().requires_grad_(True) for t in (c, w, b)]
y64 = F.conv_transpose2d(F.relu(ref(a64[0])), a64[1], a64[2], padding=2)
y64.backward(gy.double())
for rep in range(2):
    d = (outs[rep][0].double().cpu() - a64[0].grad).abs()
    per_img = d.amax(dim=(1, 2, 3))
    bad = (per_img > 1e-4 * a64[0].grad.abs().max()).nonzero().flatten()
    per_ch = d.amax(dim=(0, 2, 3))
    print('fused run %d: gc err %.1e, wrong images %d %s; per-channel max err %s' % (rep, rel(outs[rep][0], a64[0].grad), bad.numel(), bad[:24].tolist(),
          ['%.0e' % v for v in per_ch.tolist()]))
    print('   ggamma %.1e gbeta %.1e' % (rel(outs[rep][1], ref.weight.grad), rel(outs[rep][2], ref.bias.grad)))
# 3. the d/d-input convolution alone on the same gy
ga = torch.empty(B, C, H, H, device='cuda')
from vae_gp_ode_amd import _lib
from vae_gp_ode_amd.ops import _ptr, _stream
_lib.call('gpode_conv2d_fwd', _ptr(gy.cuda()), _ptr(w.cuda()), _ptr(None), _ptr(ga), B, 1, 28, 28, 16, 5, 1, 2, 28, 28, _stream())
ga64 = F.conv2d(gy.double(), w.double(), padding=2)
print('d/d-input conv alone: %.1e' % rel(ga, ga64))
