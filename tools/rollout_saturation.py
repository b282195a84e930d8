"""Rollout launch time and fraction of the fp32 vector peak against the number of trajectories per launch (the benchmark
configurations hold 32 - 256 per GPU: one 4-wavefront team per CU or less).  usage: python tools/rollout_saturation.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench as B
from vae_gp_ode_amd import ops

dev = torch.device('cuda', 0)
for name in ('cfg1', 'cfg2', 'cfg3'):
    w0 = B.WORKLOADS[name]
    for batch in (32, 256, 1024, 4096, 16384):
        w = dict(w0, batch=batch)
        flow, p, nz, z0, ts, nzd, z0d, tsd = B.make_inputs(w, 121, dev, 0)
        gp = flow.odefunc.diffeq
        with torch.no_grad():
            c = gp.build_cache(noise=nzd)
            for _ in range(3):
                ops.rollout(c, z0d, tsd, w['order'], 'rk4')
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            for e0, e1 in evs:
                e0.record(); ops.rollout(c, z0d, tsd, w['order'], 'rk4'); e1.record()
            torch.cuda.synchronize()
        ms = float(np.median([a.elapsed_time(b) for a, b in evs]))
        tf = w['mflop'] * 1e6 * batch / (ms * 1e-3) / 1e12
        print('%s  %6d trajectories  rollout %8.3f ms  %7.2f TFLOP/s  frac %.3f  %.3g trajectories/s' %
              (name, batch, ms, tf, tf / B.PEAK_FP32_TFLOPS, batch / (ms * 1e-3)), flush=True)
