#!/usr/bin/env python3
"""bench.py -- latent trajectories/s of the GP-ODE hot path on MI355X (driver contract).

    python bench.py --gpus N --steps K --warmup W [--workload cfg2|cfg4|cfg3|cfg1]

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM.
  --mode elbo (default): the full training step of main.py:199-211 -- encoder, GP draw (cache build), RK4
      rollout, decoder, ELBO, loss.backward(), Adam -- every kernel hand-written HIP, L = 1.
  --mode integrator: GP draw + RK4 rollout only (forward).
One process per GPU; for N>1 the minibatch axis is sharded (every rank owns `batch` sequences, all ranks
integrate under the SAME GP draw) and, in elbo mode, the gradient bucket is all-reduced over RCCL before the
optimizer step -> weak scaling.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel, timed live with HIP events on the
launch stream; `cpu_baseline` times the CPU oracle (kind "port": the reference itself never travels to the
GPU box) on a bounded sample of the same workload.
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

# BASELINE.json configs; flops per trajectory from SURVEY.md section 8(d): mflop = integrator (rk4), elbo_mflop = full ELBO step
# (forward + backward, the 3x rule)
WORKLOADS = {
    # name: kernel, order, q, M, S, T, batch per GPU, integrator MFLOP/traj
    'cfg1': dict(kernel='RBF', order=1, q=6, M=100, S=256, T=16, batch=32, mflop=2.252, elbo_mflop=452.0,
                 desc='configs[0]: ode1 RBF q=6 M=100 S=256 T=16 batch=32'),
    'cfg2': dict(kernel='DF', order=1, q=6, M=100, S=256, T=16, batch=256, mflop=5.864, elbo_mflop=463.0,
                 desc='configs[1]: ode1 DF q=6 M=100 S=256 T=16 batch=256'),
    'cfg3': dict(kernel='RBF', order=2, q=3, M=100, S=256, T=16, batch=256, mflop=1.127, elbo_mflop=452.0,
                 desc='configs[2]: ode2 RBF Din=6 Dout=3 M=100 S=256 T=16 batch=256'),
    'cfg4': dict(kernel='RBF', order=1, q=6, M=100, S=256, T=16, batch=256, mflop=2.252, elbo_mflop=452.0,
                 desc='configs[3] per-GPU shard: ode1 RBF q=6 M=100 S=256 T=16 batch=256/GPU (2048 over 8)'),
    # K_uu is 8192 x 8192; forward and backward take the streamed 4-wavefront team (pack read from L2)
    'cfg5': dict(kernel='DF', order=1, q=16, M=512, S=256, T=64, batch=128, mflop=438.7, elbo_mflop=3094.0,
                 desc='configs[4] per-GPU shard: ode1 DF q=16 M=512 S=256 T=64 batch=128/GPU (1024 over 8)'),
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


def make_model_inputs(w, seed, dev, rank):
    """Full model of the reference's shape (build_model under seed_everything, README hyper-parameters) and a
    synthetic normalised minibatch X (N,T,1,28,28) (data/utils.py:8-15 on random frames)."""
    import types
    from vae_gp_ode_amd.model.core.initialization import initialize_and_fix_kernel_parameters
    from vae_gp_ode_amd.model.core.noise import install_device_noise
    from vae_gp_ode_amd.model.create_model import build_model
    from vae_gp_ode_amd.model.misc.torch_utils import seed_everything
    q, order = w['q'], w['order']
    args = types.SimpleNamespace(D_in=q * order, D_out=q, num_inducing=w['M'], num_features=w['S'], dimwise=True, q_diag=False,
                                 device=dev, kernel=w['kernel'], ode=order, solver='rk4', use_adjoint=False, frames=5, n_filt=8,
                                 latent_dim=q, Ndata=360, dt=0.1)
    seed_everything(seed)
    model = build_model(args).to(dev)
    initialize_and_fix_kernel_parameters(model, lengthscale_value=2.0, variance_value=1.0)
    # the function draw is the same on every rank; the encoders' reparameterisation noise is the rank's own (one rank: all of it in one launch)
    install_device_noise(model, seed + 1, eps_seed=None if int(os.environ.get('WORLD_SIZE', '1')) == 1 else seed + 7919 * (rank + 1))
    # Keep the synthetic run finite: with the reference's own init, a batch of 4096 random-latent images puts a few
    # sigmoid outputs at exactly 1.0f, and log(1 - z) = -inf poisons the ELBO (the reference's NaN guard exists
    # for this).  Shrinking the last decoder layer keeps every pixel off the fp32 saturation point;
    # the work per step is unchanged.
    with torch.no_grad():
        model.vae.decoder.decnn[10].weight.mul_(0.25)
    gx = torch.Generator().manual_seed(seed + 100 + rank)
    X = ((torch.rand(w['batch'], w['T'], 1, 28, 28, generator=gx) - 0.1307) / 0.3081)
    return model, X


def cpu_baseline_elbo(w, model, X, budget_s=15.0):
    """The same training step on the CPU: oracle compute_loss (torch CPU) + autograd + torch.optim.Adam."""
    from oracle import gpode_oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get('BENCH_CPU_THREADS', '16'))))
    torch.set_num_threads(cores)
    sd = {k: (v.detach().cpu().clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k and '_num_evals' not in k
              else v.detach().cpu().clone()) for k, v in model.state_dict().items()}
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-6)          # as the GPU leg: the same work at any learning rate, and the state stays where it was
    q, order = w['q'], w['order']
    Di, Do, M, S, N = q * order, q, w['M'], w['S'], X.shape[0]
    g = torch.Generator().manual_seed(7)

    def step():
        nz = dict(eps_u=torch.randn(M, Do, generator=g), rff_w=torch.randn(S if w['kernel'] == 'RBF' else 2 * S, Do, generator=g),
                  rff_eps=torch.randn(Di, S, Do, generator=g), rff_u=torch.rand(1, S, Do, generator=g))
        eps_s = torch.randn(N, q, generator=g)
        eps_v = torch.randn(N, q, generator=g) if order == 2 else None
        opt.zero_grad()
        r = O.compute_loss(X, sd, [nz], eps_s, eps_v, kernel=w['kernel'], order=order, method='rk4', dt=0.1, Ndata=360)
        r['loss'].backward()
        # one fp32 sigmoid output at exactly 1.0 makes log(1 - z) = -inf (the reference's training loop exits there,
        # main.py:205-207); the timed work of such a step is done, its update is dropped so that the next step can run
        if torch.isfinite(r['loss']):
            opt.step()
    step()
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 200:
            break
    return dict(value=N * n / el, unit='trajectories/s', cores=cores, kind='port',
                sample='%d full training steps (fwd + bwd + Adam, batch %d) of the torch-CPU oracle in %.1f s' % (n, N, el))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes of this script (one per GPU, rendezvous on
    127.0.0.1), relay rank 0's JSON line, return the worst exit code.  Runs BEFORE anything touches the GPU in this process, and
    the ranks are children -- a process that has initialised the GPU is never replaced."""
    import signal
    import subprocess
    env = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
               MASTER_PORT=os.environ.get('MASTER_PORT') or str(_free_port()))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')    # dmabuf IPC: RCCL across processes needs it on this driver
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        # rank 0 inherits stdout (its JSON line is the job's line); the other ranks' stdout goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e,
                                      stdout=None if r == 0 else sys.stderr))
    worst = 0
    alive = set(range(n))
    try:
        while alive:
            for r in sorted(alive):
                rc = procs[r].poll()
                if rc is None:
                    continue
                alive.discard(r)
                if rc != 0:
                    worst = worst or rc
                    print('[bench] rank %d exited with code %d; stopping the other ranks' % (r, rc), file=sys.stderr, flush=True)
                    for o in sorted(alive):              # exactly the PIDs started above
                        procs[o].send_signal(signal.SIGTERM)
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return worst


def run_dry(a, rank, world, dist):
    """GPODE_BENCH_DRYRUN=1: the launcher / rendezvous / timing-protocol plumbing of the N > 1 path with an empty step, over
    gloo on the CPU (tests/test_bench_launcher.py).  Not a measurement and labelled as such."""
    def barrier():
        if dist is not None:
            dist.barrier()
    for _ in range(a.warmup):
        pass
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        time.sleep(1e-3)
    barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = t.item()
    w = WORKLOADS[a.workload]
    return {'metric': 'dry_run', 'value': w['batch'] * world * a.steps / el, 'unit': 'trajectories/s', 'n_gpus': world,
            'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': el / a.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'none (dry run of the launcher: empty step, gloo on the CPU)',
            'config': {'workload': 'dry run', 'parallelism': 'dp%d' % world}, 'dry_run': True,
            'dist_backend': dist.get_backend() if dist is not None else None,
            'dist_ranks': dist.get_world_size() if dist is not None else 1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--workload', default='cfg2', choices=sorted(WORKLOADS))
    ap.add_argument('--mode', default='elbo', choices=['elbo', 'integrator'])
    ap.add_argument('--seed', type=int, default=121)
    ap.add_argument('--L', type=int, default=1, help='Monte-Carlo draws per step (main.py:200 trains at L = 1, then L = 5); a trajectory '
                    'is one (sample, draw) pair, so a step integrates batch x L of them')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip extra.configs (configs[0], configs[2], integrator-only figures)')
    ap.add_argument('--no-overlap', action='store_true', help='keep the GP cache build / cache backward on the main stream')
    ap.add_argument('--no-sync-bn', action='store_true', help='N > 1: BatchNorm with per-rank statistics instead of the global minibatch')
    ap.add_argument('--dp-graph', default='whole', choices=['whole', 'fwdbwd', 'off'],
                    help='N > 1 over RCCL: what one HIP graph holds (whole step incl. collectives / forward+backward / nothing)')
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel of the step eagerly instead of replaying a captured HIP graph')
    a = ap.parse_args()

    # --gpus N without a launcher (the driver's `python bench.py --gpus N`): this process only starts the N ranks.  Nothing
    # above or inside spawn_ranks touches the GPU.
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(spawn_ranks(a.gpus, sys.argv[1:]))
    # the job's stdout carries ONE JSON line: keep a private handle to it and point fd 1 at stderr, so that whatever the
    # communication libraries print on stdout (gloo's "[Gloo] Rank 0 is connected ...") cannot end up in front of that line
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (a.gpus, world))
    if os.environ.get('GPODE_BENCH_DRYRUN') == '1':
        if os.environ.get('GPODE_BENCH_DRYRUN_FAIL_RANK') == str(rank):     # tests: a rank that dies before the rendezvous
            raise SystemExit(3)
        dist = None
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group('gloo')
        out = run_dry(a, rank, world, dist)
        if rank == 0:
            print(json.dumps(out), file=json_out, flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    # GPODE_BENCH_REHEARSE=1: all ranks share the visible card(s) and talk over gloo -- a rehearsal of the N > 1 code path
    # (sharded minibatch, gradient bucket, all-reduce between graph replays) on a one-GPU box; not a measurement
    rehearse = os.environ.get('GPODE_BENCH_REHEARSE') == '1'
    ndev = torch.cuda.device_count()
    if rehearse:
        local = local % ndev
    elif local >= ndev:
        raise SystemExit('bench.py: rank %d needs cuda:%d but only %d device(s) are visible (one process per GPU)' % (rank, local, ndev))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    # GPODE_BENCH_FORCE_DIST=1: take the distributed (RCCL) branch even with one rank -- proves the nccl code path on a one-GPU box
    if world > 1 or os.environ.get('GPODE_BENCH_FORCE_DIST') == '1':
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', str(_free_port()))
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        if rehearse:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)
    n_gpus = world
    w = WORKLOADS[a.workload]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if a.mode == 'integrator':
        out = run_integrator(a, w, dev, rank, n_gpus, dist, barrier)
    else:
        out = run_elbo(a, w, dev, rank, n_gpus, dist, barrier)
    if dist is not None:
        out['dist_backend'] = dist.get_backend()
        out['dist_ranks'] = dist.get_world_size()
        out['rccl_ranks'] = dist.get_world_size() if dist.get_backend() == 'nccl' else 0
    if out['n_gpus'] != a.gpus:
        raise SystemExit('bench.py: measured on %d rank(s) but --gpus %d was asked for' % (out['n_gpus'], a.gpus))
    if rank == 0:
        print(json.dumps(out), file=json_out, flush=True)
    if dist is not None:
        dist.destroy_process_group()


def run_elbo(a, w, dev, rank, n_gpus, dist, barrier):
    from vae_gp_ode_amd.model.create_model import backward, compute_loss
    from vae_gp_ode_amd.optim import HipAdam
    from vae_gp_ode_amd.parallel import GradAllReduce
    model, X = make_model_inputs(w, a.seed, dev, rank)
    Xd = X.to(dev)
    # lr 1e-6: the step does the same work at any learning rate, and a benchmark must finish whatever --steps is.  With the
    # reference's lr = 1e-3 = initial diag(Us_sqrt), Adam's first step lands some diagonal entries on exactly 0 (log 0 in the
    # inducing KL); and on random-noise targets normalised to [-0.42, 2.8] (data/utils.py:8-15) the Bernoulli likelihood is
    # unbounded, so any learning rate that moves the decoder drives sigmoid outputs to exactly 1.0f within a few
    # thousand steps (lr 1e-4: non-finite loss before step 5000).  Adam moves a parameter by <= lr per step: 1e-6 keeps 1e5 steps
    # within 0.1 of the initial state.
    # one GPU: no gradient bucket; N > 1: the bucket is filled in one launch from the produced gradients before the all-reduce
    opt = HipAdam(model.parameters(), lr=1e-6, bucketed='gather' if dist is not None else False)
    sync = GradAllReduce(opt.flat_grads, dist) if dist is not None else None
    bn_sync = None
    if dist is not None and not a.no_sync_bn:          # BatchNorm over the GLOBAL minibatch, as the single-process reference
        from vae_gp_ode_amd import vae_ops
        from vae_gp_ode_amd.parallel import BatchNormSync
        bn_sync = BatchNormSync(dist)
        vae_ops.set_bn_sync(bn_sync)
    last = {}

    from vae_gp_ode_amd import ops
    ops.set_overlap(not a.no_overlap)   # GP cache build / cache backward on a side stream, next to the encoder's kernels

    def fwd_bwd():
        opt.zero_grad()
        loss, nl, klr, klu = compute_loss(model, Xd, a.L)
        backward(loss)
        ops.join_side_stream()
        return loss

    def eager_step():
        last['loss'] = fwd_bwd()
        if sync is not None:
            sync.all_reduce_grads()
        opt.step()

    def whole_step():
        loss = fwd_bwd()
        opt.step()
        return loss

    def dp_step():
        loss = fwd_bwd()
        sync.all_reduce_grads()
        opt.step()
        return loss

    # One HIP graph per step instead of ~250 launches (vae_gp_ode_amd/graph.py).
    #   N = 1: the whole step.
    #   N > 1 over RCCL: the whole step as well -- the BatchNorm statistics all-gathers, the gradient all-reduce and the Adam
    #     launch are stream operations on the captured stream and become nodes of the same graph (--dp-graph fwdbwd keeps the
    #     gradient all-reduce and Adam eager between replays; off launches everything eagerly).
    #   N > 1 over gloo (rehearsal on one card): gloo's collectives are host code and cannot be captured -- forward + backward
    #     are replayed when they hold no collective (--no-sync-bn), otherwise the step runs eagerly.
    # what one step asks of the interconnect: count and payload of its collectives.  Counted while the step function runs as Python
    # anyway (GraphedStep's warm-up steps and its capture pass, or the first eager step) -- NOT in an eager step of its own on the
    # default stream ahead of the capture: that pins autograd's AccumulateGrad nodes to the default stream, the captured backward
    # then synchronises with it, and hipStreamEndCapture dies on the unjoined stream (a segmentation fault, not an error).
    coll = None
    n_counted = [0]
    if sync is not None:
        from vae_gp_ode_amd import parallel
        parallel.stats.per_step, parallel.stats.on = {}, True

    def counted(fn):
        def run():
            n_counted[0] += 1
            return fn()
        return run if sync is not None else fn
    step, graphed, mode = eager_step, False, 'eager'
    if not a.no_graph:
        if sync is None:
            mode = 'whole'
        elif dist.get_backend() == 'nccl':
            mode = a.dp_graph
        else:
            mode = 'fwdbwd' if bn_sync is None and a.dp_graph != 'off' else 'off'
    if mode in ('whole', 'fwdbwd'):
        try:
            from vae_gp_ode_amd.graph import GraphedStep, device_generators
            gens = device_generators(model)
            if mode == 'whole':
                g = GraphedStep(counted(whole_step if sync is None else dp_step), generators=gens, warmup=2)

                def step():
                    last['loss'] = g()
            else:
                g = GraphedStep(counted(fwd_bwd), generators=gens, warmup=2, grad_params=opt.params)

                def step():
                    last['loss'] = g()
                    sync.all_reduce_grads()
                    opt.step()
            graphed = True
        except Exception as e:  # capture is an optimisation of the launch path, not of the kernels: report and run eagerly
            import traceback
            traceback.print_exc(limit=14, file=sys.stderr)
            print('[bench] HIP graph capture failed (%s); running the step eagerly' % type(e).__name__, file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            step, mode = eager_step, 'eager (capture failed)'
    if sync is not None:
        from vae_gp_ode_amd import parallel
        if n_counted[0] == 0:                        # no graph: the first eager step is the counted one
            eager_step()
            n_counted[0] = 1
        parallel.stats.on = False
        coll = {k: {'count': v['count'] // n_counted[0], 'payload_bytes': v['payload_bytes'] // n_counted[0]}
                for k, v in parallel.stats.summary().items()}
        if mode == 'fwdbwd':                         # the gradient all-reduce stays between the replays: one per step
            coll.setdefault('all_reduce', {'count': 1, 'payload_bytes': int(opt.flat_grads.flat.numel() * 4)})
    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    el = time.perf_counter() - t0
    if not torch.isfinite(last['loss']).all():
        raise SystemExit('non-finite loss')
    if dist is not None:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = t.item()
    value = w['batch'] * a.L * n_gpus * a.steps / el          # trajectories = (sample, Monte-Carlo draw) pairs
    roof = dominant_kernel_roofline(w, dev, a.L)
    out = {
        'metric': 'latent_trajectories_per_sec', 'value': value, 'unit': 'trajectories/s',
        'n_gpus': n_gpus, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': el / a.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': w['desc'] + '; step = full ELBO training step (encoder, GP draw, rk4 rollout, decoder, ELBO, backward, Adam), L=%d' % a.L,
                   'global_batch': w['batch'] * n_gpus, 'mc_draws': a.L, 'solver': 'rk4 (3/8 rule)', 'parallelism': 'dp%d' % n_gpus,
                   'hip_graph': graphed, 'graph_scope': mode, 'gp_side_stream': not a.no_overlap,
                   'batchnorm': 'global-minibatch statistics (all-gather per layer)' if bn_sync is not None else 'per-process statistics'},
        'elbo_step_ms': el / a.steps * 1e3,
        'roofline': roof,
    }
    if coll is not None:
        out['collectives_per_step'] = coll
    # whole-step roofline: SURVEY 8(d)'s algorithmic flops per trajectory (forward + backward) x trajectories/s over the fp32 peak
    roof['step_frac'] = w['elbo_mflop'] * 1e6 * (value / n_gpus) / 1e12 / PEAK_FP32_TFLOPS
    roof['step_note'] = '%.0f MFLOP per trajectory (SURVEY 8d, full ELBO step) x per-GPU trajectories/s / %.1f TFLOP/s' % (w['elbo_mflop'], PEAK_FP32_TFLOPS)
    if rank == 0:
        print('[bench] gpu leg done: %.1f traj/s, %.3f ms/step' % (value, el / a.steps * 1e3), file=sys.stderr, flush=True)
        if n_gpus == 1 and not a.no_extra:
            extra = other_configs(a, dev, a.workload)      # (same launch form as the headline leg: graph replay, side-stream overlap)
            main = extra.pop(a.workload)
            if 'L5' in main:
                if a.L == 1:
                    main['L5']['elbo_step_ms_over_L1'] = main['L5']['elbo_step_ms'] / (el / a.steps * 1e3)
                out['L5'] = main['L5']
            out['integrator_ms'] = main['integrator_ms']
            out['integrator_traj_per_s'] = main['integrator_traj_per_s']
            roof['integrator'] = {'bound': 'valu', 'kernel': 'rollout_team_kernel (rk4 rollout of the GP-ODE, one launch per draw)',
                                  'achieved': main['rollout_valu_frac'] * PEAK_FP32_TFLOPS, 'peak': PEAK_FP32_TFLOPS, 'unit': 'TFLOP/s',
                                  'frac': main['rollout_valu_frac'], 'ms_per_launch': main['rollout_ms'],
                                  'note': '%.3f MFLOP/trajectory x %d trajectories (SURVEY 8d; transcendentals count as 1 flop)' % (w['mflop'], w['batch']),
                                  'saturated': main['rollout_saturated']}
            out['extra'] = {'configs': extra}
        if not a.no_cpu_baseline and n_gpus == 1:
            out['cpu_baseline'] = cpu_baseline_elbo(w, model, X)
            out['gpu_over_cpu'] = value / out['cpu_baseline']['value']
    return out


def quick_integrator(w, dev, seed, steps=50, warmup=5, L=1):
    """GP draw + rk4 rollout of workload `w` on one GPU: (ms per graph-replayed draw + rollout, ms of the rollout launch alone
    bracketed by events on the launch stream, the inputs for a CPU leg).  L > 1: L draws through ONE build + ONE rollout launch."""
    from vae_gp_ode_amd import ops
    from vae_gp_ode_amd.graph import GraphedStep
    flow, p, nz, z0, ts, nzd, z0d, tsd = make_inputs(w, seed, dev, 0)
    gp = flow.odefunc.diffeq
    if L > 1:                                        # L different function draws (a leading draw axis on every noise tensor)
        gl = torch.Generator().manual_seed(seed + 2)
        nzL = {k: torch.stack([v] + [torch.rand(v.shape, generator=gl) if k == 'rff_u' else torch.randn(v.shape, generator=gl)
                                     for _ in range(L - 1)]).to(dev) for k, v in nz.items()}
    with torch.no_grad():
        def step(ev=None):
            if L > 1:
                c = gp.build_cache(noise=nzL, draws=L)
            else:
                c = gp.build_cache(noise=nzd)
            if ev is not None:
                ev[0].record()
            zt = ops.rollout(c, z0d, tsd, w['order'], 'rk4')
            if ev is not None:
                ev[1].record()
            return zt
        for _ in range(3):
            step()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for ev in evs:
            step(ev)
        torch.cuda.synchronize()
        roll_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in evs]))
        g = GraphedStep(step, warmup=1)
        for _ in range(warmup):
            g()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            zt = g()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    if not torch.isfinite(zt).all():
        raise SystemExit('non-finite trajectories (%s)' % w['desc'])
    return ms, roll_ms, (p, nz, z0, ts)


def quick_elbo(w, dev, seed, steps=30, warmup=5, L=1):
    """Full ELBO training step of workload `w` on one GPU, graph-replayed (the launch form of the headline leg: side-stream
    overlap as set by the caller): (ms per step, model, X)."""
    from vae_gp_ode_amd import ops
    from vae_gp_ode_amd.graph import GraphedStep, device_generators
    from vae_gp_ode_amd.model.create_model import backward, compute_loss
    from vae_gp_ode_amd.optim import HipAdam
    model, X = make_model_inputs(w, seed, dev, 0)
    Xd = X.to(dev)
    opt = HipAdam(model.parameters(), lr=1e-6, bucketed=False)

    def whole_step():
        opt.zero_grad()
        loss, *_ = compute_loss(model, Xd, L)
        backward(loss)
        ops.join_side_stream()
        opt.step()
        return loss
    g = GraphedStep(whole_step, generators=device_generators(model), warmup=2)
    for _ in range(warmup):
        g()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = g()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    if not torch.isfinite(loss).all():
        raise SystemExit('non-finite loss (%s)' % w['desc'])
    return ms, model, X


def rollout_only_ms(w, dev, seed, reps=10):
    """Median time of the rollout launch alone (HIP events on the launch stream) for workload `w`."""
    from vae_gp_ode_amd import ops
    flow, p, nz, z0, ts, nzd, z0d, tsd = make_inputs(w, seed, dev, 0)
    gp = flow.odefunc.diffeq
    with torch.no_grad():
        c = gp.build_cache(noise=nzd)
        for _ in range(3):
            ops.rollout(c, z0d, tsd, w['order'], 'rk4')
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for e0, e1 in evs:
            e0.record()
            ops.rollout(c, z0d, tsd, w['order'], 'rk4')
            e1.record()
        torch.cuda.synchronize()
    return float(np.median([e0.elapsed_time(e1) for e0, e1 in evs]))


def other_configs(a, dev, main_name):
    """`extra.configs` of the line, one GPU: every BASELINE configuration gets a driver-run number -- configs[0] (the reference's
    CPU-runnable case and the north-star configuration, with the CPU oracle timed beside it in both modes and the L = 5 form of
    main.py:200), configs[2], the per-GPU shards of configs[3] and configs[4] (no CPU legs), plus the integrator-only figures of the
    headline workload."""
    out = {}
    for name in dict.fromkeys(('cfg1', 'cfg3', 'cfg4', 'cfg5', main_name)):
        w = WORKLOADS[name]
        rec = {'workload': w['desc']}
        heavy = name == 'cfg5'                       # 48 ms steps: fewer repetitions, no saturation sweep
        ims, roll_ms, cpu_in = quick_integrator(w, dev, a.seed, steps=10 if heavy else 50)
        rec['integrator_ms'] = ims
        rec['integrator_traj_per_s'] = w['batch'] / (ims * 1e-3)
        rec['rollout_ms'] = roll_ms
        rec['rollout_valu_frac'] = w['mflop'] * 1e6 * w['batch'] / (roll_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS
        if not heavy:
            # the same rollout kernel with 16 x the trajectories per launch: the benchmark batches (32 - 256 per GPU) are one 4-wavefront
            # team per CU or less, i.e. on the kernel's latency floor; this is where it saturates (tools/rollout_saturation.py)
            big = rollout_only_ms(dict(w, batch=16 * max(w['batch'], 256)), dev, a.seed)
            rec['rollout_saturated'] = {'trajectories': 16 * max(w['batch'], 256), 'rollout_ms': big,
                                        'valu_frac': w['mflop'] * 1e6 * 16 * max(w['batch'], 256) / (big * 1e-3) / 1e12 / PEAK_FP32_TFLOPS}
        if name != main_name:
            ems, model, X = quick_elbo(w, dev, a.seed, steps=8 if heavy else 30, warmup=2 if heavy else 5)
            rec['elbo_step_ms'] = ems
            rec['elbo_traj_per_s'] = w['batch'] / (ems * 1e-3)
            rec['elbo_step_frac'] = w['elbo_mflop'] * 1e6 * rec['elbo_traj_per_s'] / 1e12 / PEAK_FP32_TFLOPS
        if name in ('cfg1', main_name) and not heavy:
            # L = 5 (main.py:200): ONE cache build that factors K_uu once, ONE rollout launch over 5 x batch trajectories, one
            # reverse sweep, one cache backward -- the reference runs five of each
            L = 5
            ims5, roll5, _ = quick_integrator(w, dev, a.seed, L=L)
            ems5, _, _ = quick_elbo(w, dev, a.seed, L=L)
            l1_step = rec.get('elbo_step_ms')
            rec['L5'] = {'mc_draws': L, 'integrator_ms': ims5, 'integrator_traj_per_s': L * w['batch'] / (ims5 * 1e-3), 'rollout_ms': roll5,
                         'integrator_ms_over_L1': ims5 / ims, 'elbo_step_ms': ems5, 'elbo_traj_per_s': L * w['batch'] / (ems5 * 1e-3),
                         'elbo_step_ms_over_L1': (ems5 / l1_step) if l1_step else None,
                         'elbo_step_frac': w['elbo_mflop'] * 1e6 * L * w['batch'] / (ems5 * 1e-3) / 1e12 / PEAK_FP32_TFLOPS}
        if name == 'cfg1' and not a.no_cpu_baseline:
            rec['cpu_baseline_integrator'] = cpu_baseline(w, *cpu_in, budget_s=4.0)
            rec['integrator_gpu_over_cpu'] = rec['integrator_traj_per_s'] / rec['cpu_baseline_integrator']['value']
            if name != main_name:                    # (the headline workload's own ELBO step has its CPU leg in the main line)
                rec['cpu_baseline_elbo'] = cpu_baseline_elbo(w, model, X, budget_s=6.0)
                rec['elbo_gpu_over_cpu'] = rec['elbo_traj_per_s'] / rec['cpu_baseline_elbo']['value']
        out[name] = rec
        print('[bench] %s: %s' % (name, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in rec.items() if not isinstance(v, str)}),
              file=sys.stderr, flush=True)
    return out


# the six matrix-core convolution kernels that carry the step's FLOPs: (layer, Cin, Cout, Hi, Ht, MMAC per image)
CONV_LAYERS = (('decnn.7', 32, 16, 13, 28, 2.163e6), ('decnn.4', 64, 32, 6, 13, 1.843e6))


def conv_kernel_launchers(B, dev):
    """name -> (launch(), algorithmic flops, algorithmic HBM bytes) for the forward (BatchNorm + ReLU folded into the input
    staging, as the training step runs it), d/d input and d/d weight of decnn.7 and decnn.4 at B images, through the C ABI."""
    import ctypes  # noqa: F401
    from vae_gp_ode_amd import _lib
    from vae_gp_ode_amd.ops import _ptr, _stream
    lib = _lib.load()
    out = {}
    for name, Cin, Cout, Hi, Ht, macs in CONV_LAYERS:
        K, S, P = 5, 2, 1
        c = torch.randn(B, Cin, Hi, Hi, device=dev)
        y = torch.empty(B, Cout, Ht, Ht, device=dev)
        gy = torch.randn(B, Cout, Ht, Ht, device=dev)
        gc = torch.empty(B, Cin, Hi, Hi, device=dev)
        wt = torch.randn(Cin, Cout, K, K, device=dev) * 0.05
        bias = torch.zeros(Cout, device=dev)
        table = torch.rand(Cin, 4, device=dev) + 0.5
        gw = torch.empty(Cin, Cout, K, K, device=dev)
        ws = torch.empty(max(int(lib.gpode_conv_wgrad_scratch(B, Cout, Cin, K)), 4), device=dev)
        geo = (B, Cout, Ht, Ht, Cin, K, S, P, Hi, Hi)
        nin, nout, nw = 4 * B * Cin * Hi * Hi, 4 * B * Cout * Ht * Ht, 4 * Cin * Cout * K * K
        fl = 2.0 * macs * B
        keep = (c, y, gy, gc, wt, bias, table, gw, ws)
        out[name + ' forward'] = (lambda c=c, table=table, wt=wt, bias=bias, y=y, geo=geo, keep=keep: _lib.call(
            'gpode_conv2d_bwd_data_bn', _ptr(c), _ptr(table), _ptr(wt), _ptr(bias), _ptr(y), *geo, _stream()), fl, nin + nout + nw)
        out[name + ' d/d input'] = (lambda gy=gy, wt=wt, gc=gc, geo=geo, keep=keep: _lib.call(
            'gpode_conv2d_fwd', _ptr(gy), _ptr(wt), _ptr(None), _ptr(gc), *geo, _stream()), fl, nin + nout + nw)
        out[name + ' d/d weight'] = (lambda gy=gy, c=c, table=table, gw=gw, ws=ws, geo=geo, keep=keep: _lib.call(
            'gpode_conv2d_bwd_weight_bn', _ptr(gy), _ptr(c), _ptr(table), _ptr(gw), _ptr(None), _ptr(ws), *geo, _stream()), fl, nin + nout + nw)
    return out


def dominant_kernel_roofline(w, dev, L=1, reps=20):
    """The step's FLOPs sit in six matrix-core convolution launches (decnn.7 and decnn.4: forward, d/d input, d/d weight).  Each is
    timed alone with HIP events (torch's current stream is the launch stream); `roofline` is the LONGEST of them -- the kernel
    that bounds the step -- and `roofline.kernels` lists all six."""
    B = w['batch'] * w['T'] * L
    recs = {}
    with torch.no_grad():
        for name, (launch, flops, nbytes) in conv_kernel_launchers(B, dev).items():
            for _ in range(3):
                launch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                launch()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            ach = flops / (ms * 1e-3) / 1e12
            recs[name] = {'ms_per_launch': ms, 'achieved': ach, 'frac': ach / PEAK_FP32_TFLOPS, 'algorithmic_flops': flops,
                          'algorithmic_bytes': nbytes}
    worst = max(recs, key=lambda k: recs[k]['ms_per_launch'])
    r = recs[worst]
    # HBM bytes and matrix-pipe busy cycles per launch come from separate rocprofv3 --pmc passes (tools/gpu_pmc_kernel.sh), which
    # cannot run inside this process; the committed measurement of THIS kernel is replayed when it was taken at the same image count
    traffic = tsrc = mfma_busy = msrc = None
    try:
        here = os.path.dirname(os.path.abspath(__file__))
        rec = json.load(open(os.path.join(here, 'profiles', 'r03_roofline_pmc.json'))).get(worst)
        if rec and rec.get('images') == B:
            traffic, tsrc = rec['hbm_bytes_per_launch'], ('profiles/r03_roofline_pmc.json: replayed from the committed rocprofv3 --pmc passes of this build '
                                                          '(FETCH_SIZE x2 + WRITE_SIZE, separate passes; counters cannot be read from inside this process)')
            mfma_busy, msrc = rec.get('mfma_busy_frac'), 'profiles/r03_roofline_pmc.json (SQ_VALU_MFMA_BUSY_CYCLES pass)'
    except (OSError, ValueError, KeyError):
        pass
    return {'bound': 'mfma', 'kernel': worst + ' (the longest kernel of the step; v_mfma_f32_16x16x4_f32, exact fp32)',
            'achieved': r['achieved'], 'peak': PEAK_FP32_TFLOPS, 'unit': 'TFLOP/s', 'frac': r['frac'], 'traffic': traffic, 'traffic_source': tsrc,
            'mfma_busy_frac_pmc': mfma_busy, 'mfma_busy_frac_source': msrc,
            'algorithmic_bytes': r['algorithmic_bytes'], 'ms_per_launch': r['ms_per_launch'], 'kernels': recs,
            'note': 'algorithmic flops = 2 x MMAC/image (decnn.7 2.163, decnn.4 1.843) x %d images; exact fp32 on the matrix cores, priced '
                    'against the dense fp32 MFMA peak (256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz); the fp32 MFMA issues on the '
                    'SIMD\'s vector port, so 157.3 is only reachable with NO other vector instruction beside it (DESIGN 4.3)' % B}


def run_integrator(a, w, dev, rank, n_gpus, dist, barrier):
    from vae_gp_ode_amd import ops
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

    # rollout launch duration: a few eager steps bracketed by events on the launch stream
    for _ in range(max(a.warmup, 1)):
        step()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(a.steps, 20))]
    for ev in evs:
        zt = step(ev)
    torch.cuda.synchronize()
    # the timed loop replays the draw + rollout as one HIP graph (the cache build alone is ~25 dependent launches)
    run, graphed = step, False
    if not a.no_graph:
        try:
            from vae_gp_ode_amd.graph import GraphedStep
            g = GraphedStep(step, warmup=1)
            run, graphed = g, True
        except Exception as e:
            print('[bench] HIP graph capture failed (%s); running the step eagerly' % type(e).__name__, file=sys.stderr, flush=True)
            torch.cuda.synchronize()
    for _ in range(a.warmup):
        zt = run()
    barrier()
    t0 = time.perf_counter()
    for k in range(a.steps):
        zt = run()
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
                   'global_batch': w['batch'] * n_gpus, 'solver': 'rk4 (3/8 rule)', 'parallelism': 'dp%d' % n_gpus,
                   'hip_graph': graphed},
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
    return out


if __name__ == '__main__':
    main()
