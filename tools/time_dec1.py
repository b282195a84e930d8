"""decnn.1 forward / backward kernels alone at 512 and 4096 images (HIP events), new kernels vs the plane-scatter engine."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_gp_ode_amd import vae_ops as V
for B in (512, 4096):
    x = torch.randn(B, 32, 4, 4, device='cuda', requires_grad=True)
    w = (torch.randn(32, 64, 3, 3, device='cuda') * 0.05).requires_grad_(True)
    b = torch.zeros(64, device='cuda', requires_grad=True)
    gy = torch.randn(B, 64, 6, 6, device='cuda')
    def fwd():
        with torch.no_grad():
            return V.conv_transpose2d(x, w, b, 1, 0)
    def fb():
        y = V.conv_transpose2d(x, w, b, 1, 0)
        y.backward(gy)
    for name, f in (('forward', fwd), ('forward + backward', fb)):
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record(); torch.cuda.synchronize()
        print('decnn.1 %-20s %5d images  %.1f us' % (name, B, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
