# debugging aid only: each HIP VAE op vs torch.nn.functional on the GPU
import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_forward import relerr
from vae_gp_ode_amd import vae_ops as V
torch.manual_seed(0)
def check(name, f_hip, f_ref, *shapes):
    xs = [torch.randn(s, device='cuda', requires_grad=True) for s in shapes]
    y = f_hip(*xs); w = torch.randn_like(y); (y * w).sum().backward()
    g1 = [x.grad.clone() for x in xs]
    for x in xs: x.grad = None
    y2 = f_ref(*xs); (y2 * w).sum().backward()
    print('%-34s fwd %.1e  grads %s' % (name, relerr(y, y2), ' '.join('%.1e' % relerr(a, x.grad) for a, x in zip(g1, xs))))
class BN: pass
for B in (40, 129, 130, 131, 192):
    bn = torch.nn.BatchNorm2d(16).cuda()
    check('bn_relu B=%d' % B, lambda x, g, b: V._BatchNormTrain.apply(x, g, b, None, None, 0.1, 1e-5, 1),
          lambda x, g, b: F.relu(F.batch_norm(x, None, None, g, b, True, 0.1, 1e-5)), (B, 16, 28, 28), (16,), (16,))
    check('convT 32->16 k5 s2 p1 op1 B=%d' % B, lambda x, w, b: V.conv_transpose2d(x, w, b, 2, 1, 1),
          lambda x, w, b: F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1), (B, 32, 13, 13), (32, 16, 5, 5), (16,))
    check('conv 8->16 k5 s2 p2 B=%d' % B, lambda x, w, b: V.conv2d(x, w, b, 2, 2),
          lambda x, w, b: F.conv2d(x, w, b, stride=2, padding=2), (B, 8, 14, 14), (16, 8, 5, 5), (16,))
    check('linear 6->512 B=%d' % B, V.linear, F.linear, (B, 6), (512, 6), (512,))
