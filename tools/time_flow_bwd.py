#!/usr/bin/env python3
"""Time the pieces of one GP draw + rollout + full backward at a BASELINE workload's shapes (default configs[4]:
DF D=16, M=512, S=256, T=64, 128 trajectories), each piece bracketed by events on the current stream.

    python tools/time_flow_bwd.py [--workload cfg5] [--reps 3]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from vae_gp_ode_amd import ops
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='cfg5')
    ap.add_argument('--reps', type=int, default=3)
    a = ap.parse_args()
    w = bench.WORKLOADS[a.workload]
    dev = torch.device('cuda:0')
    flow, _, _, _, _, nz, z0, ts = bench.make_inputs(w, 121, dev, 0)
    gp = flow.odefunc.diffeq
    k = gp.kern
    params = (k.unconstrained_lengthscales, k.unconstrained_variance, gp.inducing_loc.optvar, gp.Um.optvar, gp.Us_sqrt.optvar)
    gw = torch.randn(z0.shape[0], ts.shape[0], z0.shape[1], device=dev)
    order, method = w['order'], 'rk4'
    times = {}

    def timed(name, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        e1.synchronize()
        times.setdefault(name, []).append(e0.elapsed_time(e1))
        return out

    for rep in range(a.reps + 1):
        with torch.no_grad():
            gp.set_noise(nz)
            c = timed('cache_build', lambda: gp.build_cache())
            zt, xs = timed('rollout_fwd', lambda: ops.rollout(c, z0, ts, order, method, save_stages=True))
            gz0, ast = timed('rollout_bwd', lambda: ops.rollout_bwd(c, xs, gw, ts, order, method))
            gpack = timed('param_grad', lambda: ops.param_grad(c, xs.reshape(-1, c.Di), ast.reshape(-1, c.Do)))
            prep = timed('cache_bwd_prepare', lambda: ops.cache_bwd_prepare(c))
            g = timed('cache_build_bwd', lambda: ops.cache_build_bwd(c, params[0].detach(), params[1].detach(), params[2].detach(),
                                                                       gpack, prepared=prep))
        c.check_factorisation()
        assert all(torch.isfinite(g[n]).all() for n in ('raw_ell', 'raw_var', 'Z', 'Um', 'Us')) and torch.isfinite(gz0).all()
        print('rep %d ok' % rep, flush=True)
    tot = 0.0
    for n, v in times.items():
        m = sum(v[1:]) / len(v[1:])
        tot += m
        print('%-20s %9.2f ms' % (n, m))
    print('%-20s %9.2f ms   (%s, batch %d, T %d)' % ('total', tot, a.workload, z0.shape[0], ts.shape[0]))


if __name__ == '__main__':
    main()
