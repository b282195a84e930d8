"""Which torch-native ops (aten::*) launch kernels inside one eager training step, and from where: one step of the bench's
`whole_step` under torch.profiler with Python stacks.  usage: python tools/step_native_ops.py [cfg1] [L]"""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg1'
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1
from vae_gp_ode_amd import ops  # noqa: E402
from vae_gp_ode_amd.model.create_model import compute_loss  # noqa: E402
from vae_gp_ode_amd.optim import HipAdam  # noqa: E402

dev = torch.device('cuda', 0)
w = bench.WORKLOADS[name]
model, X = bench.make_model_inputs(w, 121, dev, 0)
Xd = X.to(dev)
opt = HipAdam(model.parameters(), lr=1e-6, bucketed=False)
ops.set_overlap(True)


def step():
    opt.zero_grad()
    loss, *_ = compute_loss(model, Xd, L)
    from vae_gp_ode_amd.model.create_model import backward as _bw; _bw(loss)
    ops.join_side_stream()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for ev in prof.events():
    if not ev.name.startswith('aten::'):
        continue
    if any(k.device_type is not None and 'CUDA' in str(k.device_type) for k in []):
        pass
    kern = [k for k in ev.kernels] if hasattr(ev, 'kernels') else []
    if not kern or any(c.name.startswith('aten::') and getattr(c, 'kernels', None) for c in ev.cpu_children):
        continue                                   # keep the innermost aten op that owns the launch
    st = [s for s in (ev.stack or []) if 'vae-gp-ode_amd' in s or 'vae_gp_ode_amd' in s or 'bench.py' in s or 'step_native' in s]
    rows.append((ev.time_range.start, ev.name, tuple(tuple(x) for x in (ev.input_shapes or ()))[:2], [k.name[:40] for k in kern], st[:3]))
rows.sort()
for t, n, sh, kn, st in rows:
    print('%-26s %-34s %-44s %s' % (n, str(sh)[:34], ','.join(kn)[:44], ' <- '.join(x.split('/')[-1] for x in st)))
print(len(rows), 'torch-native leaf ops that launch kernels in one step')
