"""Launch the roofline kernel of bench.py (decoder decnn.7 forward, B = batch*T images) a few times, for
rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE need separate passes on gfx950) and --kernel-trace --stats."""
import sys, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vae_gp_ode_amd import vae_ops as V
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x = torch.randn(B, 32, 13, 13, device='cuda')
w = torch.randn(32, 16, 5, 5, device='cuda') * 0.05
b = torch.zeros(16, device='cuda')
with torch.no_grad():
    for _ in range(5):
        V.conv_transpose2d(x, w, b, 2, 1, 1)
torch.cuda.synchronize()
print('done')
