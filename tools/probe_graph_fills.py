"""Which kernels does torch launch AROUND a graph replay (generator bookkeeping), and can the executable graph be launched directly?"""
import ctypes
import torch
from torch.profiler import profile, ProfilerActivity

x = torch.zeros(1024, device='cuda')
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        y = x * 2 + 1
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
for keep in (False, True):
    try:
        g = torch.cuda.CUDAGraph(keep_graph=keep) if keep else torch.cuda.CUDAGraph()
    except TypeError as e:
        print('keep_graph unsupported', e)
        continue
    with torch.cuda.graph(g):
        y = x * 2 + 1
    if keep:
        g.instantiate()
    g.replay()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
    names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    print('keep_graph', keep, 'device events of 3 replays:', len(names))
    for n in names:
        print('   ', n[:110])
    print('raw handles:', hasattr(g, 'raw_cuda_graph'), hasattr(g, 'raw_cuda_graph_exec'))
    if keep and hasattr(g, 'raw_cuda_graph_exec'):
        ex = g.raw_cuda_graph_exec()
        print('exec handle', hex(ex))
        hip = ctypes.CDLL('libamdhip64.so')
        hip.hipGraphLaunch.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        st = torch.cuda.current_stream().cuda_stream
        x.zero_()
        y.zero_()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(3):
                rc = hip.hipGraphLaunch(ctypes.c_void_p(ex), ctypes.c_void_p(st))
            torch.cuda.synchronize()
        print('direct launch rc', rc, 'y[0]', float(y[0]))
        for e in prof.events():
            if e.device_type == torch.autograd.DeviceType.CUDA:
                print('   ', e.name[:110])
