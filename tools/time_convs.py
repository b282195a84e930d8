"""Time every decoder conv op (forward, d/d input + d/d weight) at the bench batch with HIP events."""
import sys, torch
sys.path.insert(0, '.')
from vae_gp_ode_amd import vae_ops as V
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
layers = [('decnn.1 ', (B, 32, 4, 4), (32, 64, 3, 3), (1, 0, 0), 295e3), ('decnn.4 ', (B, 64, 6, 6), (64, 32, 5, 5), (2, 1, 0), 1.843e6),
          ('decnn.7 ', (B, 32, 13, 13), (32, 16, 5, 5), (2, 1, 1), 2.163e6), ('decnn.10', (B, 16, 28, 28), (16, 1, 5, 5), (1, 2, 0), 313.6e3)]
def timeit(f, n=10):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, xs, ws, (s, p, op), macs in layers:
    x = torch.randn(xs, device='cuda', requires_grad=True); w = (torch.randn(ws, device='cuda') * 0.05).requires_grad_(True)
    b = torch.zeros(ws[1], device='cuda', requires_grad=True)
    with torch.no_grad():
        tf = timeit(lambda: V.conv_transpose2d(x, w, b, s, p, op))
    y = V.conv_transpose2d(x, w, b, s, p, op); gy = torch.randn_like(y)
    def bw():
        x.grad = w.grad = b.grad = None
        y.backward(gy, retain_graph=True)
    tb = timeit(bw)
    fl = 2 * macs * B
    print('%s fwd %7.3f ms (%5.1f TF)   bwd(data+weight) %7.3f ms (%5.1f TF)' % (name, tf, fl / tf / 1e9, tb, 2 * fl / tb / 1e9))
