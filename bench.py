#!/usr/bin/env python3
"""bench.py -- latent trajectories/s of the GP-ODE hot path on MI355X (driver contract).

    python bench.py --gpus N --steps K --warmup W [--workload cfg2|cfg4|cfg3|cfg1]

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM:
one GP function draw (cache build: K_uu, Cholesky, solves) + the fixed-grid RK4 rollout of the
whole minibatch.  One process per GPU; for N>1 the minibatch axis is sharded (every rank integrates
its own `batch` trajectories under the SAME GP draw), no data-path collective -> weak scaling.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the rollout), timed live with
HIP events on the launch stream; `cpu_baseline` times the CPU oracle (kind "port": the reference
itself never travels to the GPU box) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BASELINE.json configs; flops per trajectory from SURVEY.md section 8(d) (integrator, rk4, T=16)
WORKLOADS = {
    # name: kernel, order, q, M, S, T, batch per GPU, integrator MFLOP/traj
    'cfg1': dict(kernel='RBF', order=1, q=6, M=100, S=256, T=16, batch=32, mflop=2.252,
                 desc='configs[0]: ode1 RBF q=6 M=100 S=256 T=16 batch=32'),
    'cfg2': dict(kernel='DF', order=1, q=6, M=100, S=256, T=16, batch=256, mflop=5.864,
                 desc='configs[1]: ode1 DF q=6 M=100 S=256 T=16 batch=256'),
    'cfg3': dict(kernel='RBF', order=2, q=3, M=100, S=256, T=16, batch=256, mflop=1.127,
                 desc='configs[2]: ode2 RBF Din=6 Dout=3 M=100 S=256 T=16 batch=256'),
    'cfg4': dict(kernel='RBF', order=1, q=6, M=100, S=256, T=16, batch=256, mflop=2.252,
                 desc='configs[3] per-GPU shard: ode1 RBF q=6 M=100 S=256 T=16 batch=256/GPU (2048 over 8)'),
}
PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector == fp32 MFMA peak


def make_inputs(w, seed, dev, rank):
    """Synthetic state of the reference's shape: SVGP_Layer init (svpy.py:76-86) under seed_everything(seed),
    README hyper-parameters (lengthscale 2, variance 1, README.md:28), explicit noise, z0 ~ N(0,1)."""
    from vae_gp_ode_amd.model.core.flow import Flow
    from vae_gp_ode_amd.model.core.svpy import SVGP_Layer
    from vae_gp_ode_amd.model.misc.constraint_utils import invsoftplus
    from vae_gp_ode_amd.model.misc.torch_utils import seed_everything
    q, order = w['q'], w['order']
    Di, Do, M, S = q * order, q, w['M'], w['S']
    seed_everything(seed)
    gp = SVGP_Layer(Di, Do, M, S, kernel=w['kernel'], device=dev).to(dev)
    with torch.no_grad():
        gp.kern.unconstrained_lengthscales.copy_(invsoftplus(torch.full((Do, Di), 2.0)))
        gp.kern.unconstrained_variance.copy_(invsoftplus(torch.ones(Do)))
    flow = Flow(gp, order=order, solver='rk4').to(dev)
    g = torch.Generator().manual_seed(seed + 1)  # the GP draw: identical on every rank
    nz = dict(eps_u=torch.randn(M, Do, generator=g),
              rff_w=torch.randn(S if w['kernel'] == 'RBF' else 2 * S, Do, generator=g),
              rff_eps=torch.randn(Di, S, Do, generator=g), rff_u=torch.rand(1, S, Do, generator=g))
    gz = torch.Generator().manual_seed(seed + 100 + rank)  # each rank owns a different shard
    z0 = torch.randn(w['batch'], Di, generator=gz)
    ts = 0.1 * torch.arange(w['T'], dtype=torch.float)
    sd = {k: v.detach().cpu() for k, v in gp.state_dict().items()}
    p = dict(raw_ell=sd['kern.unconstrained_lengthscales'], raw_var=sd['kern.unconstrained_variance'],
             Z=sd['inducing_loc.optvar'], Um=sd['Um.optvar'], Us=sd['Us_sqrt.optvar'])
    return flow, p, nz, z0, ts, {k: v.to(dev) for k, v in nz.items()}, z0.to(dev), ts.to(dev)


def cpu_baseline(w, p, nz, z0, ts, budget_s=12.0):
    """CPU oracle (torch CPU, all host cores) on the same workload: GP draw + RK4 rollout."""
    from oracle import gpode_oracle as O
    # the box's CPU share, not the host's core count (a 1-GPU box is allotted 16 cores)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get('BENCH_CPU_THREADS', '16'))))
    torch.set_num_threads(cores)
    N = z0.shape[0]

    def step():
        c = O.build_cache(p, nz, w['kernel'])
        return O.flow_forward(z0, ts, c, w['order'], 'rk4')
    with torch.no_grad():
        step()  # warm-up
        t0 = time.perf_counter()
        n = 0
        while True:
            step()
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s or n >= 50:
                break
    return dict(value=N * n / el, unit='trajectories/s', cores=cores, kind='port',
                sample='%d full steps (GP draw + rk4 rollout, batch %d) of the torch-CPU oracle in %.1f s' % (n, N, el))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='cfg2', choices=sorted(WORKLOADS))
    ap.add_argument('--seed', type=int, default=121)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    a = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=dev)
    n_gpus = world

    from vae_gp_ode_amd import ops
    w = WORKLOADS[a.workload]
    flow, p, nz, z0, ts, nzd, z0d, tsd = make_inputs(w, a.seed, dev, rank)
    gp = flow.odefunc.diffeq
    torch.set_grad_enabled(False)

    def step(ev=None):
        # == Flow.forward(z0, ts) (flow.py:68-86) with the rollout launch bracketed by events
        c = gp.build_cache(noise=nzd)
        if ev is not None:
            ev[0].record()
        zt = ops.rollout(c, z0d, tsd, w['order'], 'rk4')
        if ev is not None:
            ev[1].record()
        return zt

    gp.set_noise(nzd)  # the public API path once (untimed) must agree with the bracketed form
    if not torch.equal(flow(z0d, tsd), step()):
        raise SystemExit('Flow.forward and the bracketed step disagree')

    for _ in range(a.warmup):
        step()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for k in range(a.steps):
        zt = step(evs[k])
    barrier()
    el = time.perf_counter() - t0
    if not torch.isfinite(zt).all():
        raise SystemExit('non-finite trajectories')
    if dist is not None:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = t.item()
    # kernels are launched on torch's current stream (ops._stream), so these events bracket the rollout kernel
    roll_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in evs]))
    total_traj = w['batch'] * n_gpus * a.steps
    value = total_traj / el
    flops_per_launch = w['mflop'] * 1e6 * w['batch']
    achieved = flops_per_launch / (roll_ms * 1e-3) / 1e12
    out = {
        'metric': 'latent_trajectories_per_sec', 'value': value, 'unit': 'trajectories/s',
        'n_gpus': n_gpus, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': el / a.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': w['desc'] + '; step = GP draw (K_uu, Cholesky, nu) + rk4 rollout, L=1 [integrator fwd]',
                   'global_batch': w['batch'] * n_gpus, 'solver': 'rk4 (3/8 rule)', 'parallelism': 'dp%d' % n_gpus},
        'roofline': {'bound': 'mfma', 'kernel': 'rollout_kernel', 'achieved': achieved, 'peak': PEAK_FP32_TFLOPS,
                     'unit': 'TFLOP/s', 'frac': achieved / PEAK_FP32_TFLOPS, 'traffic': None,
                     'ms_per_launch': roll_ms,
                     'note': 'fp32 VALU+transcendental kernel priced against the fp32 peak (vector = MFMA = 157.3 TF); '
                             'algorithmic flops = %.3f MFLOP/traj x %d traj (SURVEY 8d)' % (w['mflop'], w['batch'])},
    }
    if rank == 0:
        print('[bench] gpu leg done: %.1f traj/s, rollout %.3f ms/launch' % (value, roll_ms), file=sys.stderr, flush=True)
        if not a.no_cpu_baseline and n_gpus == 1:
            out['cpu_baseline'] = cpu_baseline(w, p, nz, z0, ts)
            out['gpu_over_cpu'] = value / out['cpu_baseline']['value']
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
