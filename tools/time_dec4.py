"""decnn.4 forward (BatchNorm + ReLU fused on its input) at 512 and 4096 images: plane-scatter engine vs GPODE_DEC4_TAPCOLS=1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_gp_ode_amd import vae_ops as V
for B in (512, 4096):
    c = torch.randn(B, 64, 6, 6, device='cuda')
    bn = torch.nn.BatchNorm2d(64).cuda()
    w = torch.randn(64, 32, 5, 5, device='cuda') * 0.05
    b = torch.zeros(32, device='cuda')
    with torch.no_grad():
        for _ in range(3):
            y = V.bn_relu_conv_transpose2d(c, bn, w, b, 2, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y = V.bn_relu_conv_transpose2d(c, bn, w, b, 2, 1)
        e1.record(); torch.cuda.synchronize()
    print('decnn.4 stage forward (stats + conv) %5d images  %.1f us  checksum %.6e' % (B, e0.elapsed_time(e1) / 10 * 1e3, float(y.double().sum())), flush=True)
