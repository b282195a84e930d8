"""Training entry point with the argument surface of the reference's experiments/main.py (flags, types and
defaults of main.py:23-114; loop contract of main.py:199-247): L = 1 for the first half of the epochs then 5,
NaN guard, per-epoch evaluation on the first test batch, ``odegpvae_mnist.pth`` checkpoint every epoch,
``--continue_training`` / ``--pretrained`` wiring, same log lines.  Every arithmetic step runs in the HIP
kernels behind ``model/``; the optimizer is the one-launch HIP Adam.

Data parallelism (no flag: launch with ``python -m torch.distributed.run --nproc-per-node N -m vae_gp_ode_amd.main ...``): one
process per GPU over RCCL, every rank takes its shard of each minibatch under the SAME GP draw (rank 0's host-RNG noise is
broadcast; ``--hip_graph`` seeds the device generators identically), one all-reduce of the flat gradient bucket per step,
rank 0 logs, evaluates and writes the checkpoint (vae_gp_ode_amd/parallel.py); training-mode BatchNorm normalises with the
statistics of the global minibatch (``--sync_bn``, default on), so an N-rank step is the reference's step on the whole minibatch.

Data: ``--task mnist`` goes through ``data/wrappers.load_data`` (``<data_root>/rot_mnist/rot-mnist.mat``, the reference's
file; the set is uploaded once and minibatches are gathered on the device) or, when that file is absent, tensors saved as
``<data_root>/rot_mnist_{train,test}.pt`` (shape (N,T,1,28,28), already z-normalised); ``--task synthetic`` generates rotating
blobs so the loop can be exercised without the dataset.  The plots are out of scope.
"""
import argparse
import logging
import os
import sys
import time
from datetime import datetime, timedelta

import torch

SOLVERS = ["dopri5", "bdf", "rk4", "midpoint", "adams", "explicit_adams", "fixed_adams"]
KERNELS = ['RBF', 'DF']

# (flag, type, default, help) -- the reference's 39 flags, in its order
FLAGS = [
    ('data_root', str, 'data/', 'Data location'), ('task', str, 'mnist', 'Experiment type'),
    ('mask', eval, True, 'select a subset of mnist data'), ('value', int, 3, 'training choice'),
    ('data_seqlen', int, 100, 'Training sequence length'), ('batch', int, 20, 'batch size'),
    ('T', int, 16, 'Number of time points'), ('Ndata', int, 360, 'Number training data points'),
    ('Ntest', int, 40, 'Number valid data points'), ('rotrand', eval, True, 'if True multiple initial rotatio angles'),
    ('latent_dim', int, 6, 'Latent space dimensionality'), ('n_filt', int, 8, 'Number of filters in the cnn'),
    ('frames', int, 5, 'Number of timesteps used for encoding velocity'),
    ('pretrained', eval, False, 'wheather to load pretrained vae'),
    ('vae_path', str, 'results/vae_31_10_2022-10:32/MNIST-VAE', 'pretrained VAE model path'),
    ('kernel', str, 'RBF', 'GP kernel'), ('num_features', int, 256, 'Number of Fourier basis functions'),
    ('num_inducing', int, 100, 'Number of inducing points for the sparse GP'),
    ('dimwise', eval, True, 'Specify separate lengthscales for every output dimension'),
    ('variance', float, 0.7, 'Initial value for rbf variance'), ('lengthscale', float, 2.0, 'Initial value for rbf lengthscale'),
    ('q_diag', eval, False, 'Diagonal posterior approximation for inducing variables'),
    ('ode', int, 1, 'order of ODE'), ('D_in', int, 6, 'ODE f(x) input dimensionality'), ('D_out', int, 6, 'ODE f(x) output dimensionality'),
    ('solver', str, 'euler', 'ODE solver for numerical integration'),
    ('ts_dense_scale', int, 2, 'Factor for making a dense integration time grid'),
    ('use_adjoint', eval, False, 'Use adjoint method for gradient computation'), ('dt', float, 0.1, 'numerical solver dt'),
    ('Nepoch', int, 5000, 'Number of gradient steps for model training'), ('lr', float, 0.001, 'Learning rate for model training'),
    ('eval_sample_size', int, 128, 'Number of posterior samples to evaluate the model predictive performance'),
    ('save', str, 'results/mnist', 'Directory name for saving all the model outputs'), ('seed', int, 121, 'Global seed for the training run'),
    ('log_freq', int, 5, 'Logging frequency while training'), ('device', str, 'cuda:0', 'place holder for device'),
    ('continue_training', eval, False, 'If set to True continoues training of a previous model'),
    ('model_path', str, 'None', 'path from where to load previous model, should be of the form results/mnist_*/*.pth'),
    ('Troll', int, 2, 'rollout'),
]
# extensions of this build (not in the reference): off by default so that a run is the reference's loop kernel by kernel
EXT_FLAGS = [
    ('hip_graph', eval, False, 'replay each training step as one captured HIP graph, GP chains on a side stream, GP noise drawn '
                               'on the device (INTEGRATION.md); the reference draws GP noise from host numpy generators'),
    ('device_noise', eval, False, 'draw the GP noise on the device (seeded identically on every rank) without --hip_graph'),
    ('backward_solves', str, 'adaptive', 'GP cache backward through the Cholesky factor: auto (triangular solves up to 192 rows, explicit '
                                         'triangular inverse beyond: fastest), always (solves up to 1216 rows: torch-grade accuracy on a '
                                         'rank-deficient K_uu), never, adaptive (auto, switched to always for the next epoch once the '
                                         "factor's pivots span more than 200x, i.e. approach the jitter floor)"),
    ('gp_side_stream', eval, False, 'GP cache build / cache backward on a side stream next to the encoder, without --hip_graph'),
    ('sync_bn', eval, True, 'data parallel: BatchNorm normalises with the statistics of the GLOBAL minibatch (all ranks), as the '
                            'single-process reference does (vae.py:55,58,113,116,119)'),
]
CHOICES = {'kernel': KERNELS, 'solver': SOLVERS, 'backward_solves': ['auto', 'always', 'never', 'adaptive']}  # default 'euler' is accepted only as a default, as in the reference


def make_parser():
    p = argparse.ArgumentParser('Learning latent dyanmics with OdeVaeGP')
    for name, typ, default, hlp in FLAGS + EXT_FLAGS:
        kw = dict(type=typ, default=default, help=hlp)
        if name in CHOICES:
            kw['choices'] = CHOICES[name]
        p.add_argument('--' + name, **kw)
    return p


class RunningAverage:
    """value + exponentially weighted average with period 10 (log_utils.py:4-70)."""

    def __init__(self, period=10):
        self.m, self.val, self.avg = 1.0 - 1.0 / period, None, None

    def update(self, v):
        self.val = v
        self.avg = v if self.avg is None else self.avg * self.m + v * (1 - self.m)


def get_logger(logpath):
    logger = logging.getLogger('gpode')
    logger.setLevel(logging.INFO)
    logger.handlers = []
    for h in (logging.FileHandler(logpath, mode='a'), logging.StreamHandler()):
        h.setLevel(logging.INFO)
        logger.addHandler(h)
    return logger


def synthetic_sequences(n, T, seed):
    """Rotating two-blob images, z-normalised like data/utils.py:8-15."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(28.), torch.arange(28.), indexing='ij')
    phase = torch.rand(n, 1, generator=g) * 6.2832
    ang = phase + torch.arange(T).float()[None] * (6.2832 / T)
    imgs = torch.zeros(n, T, 1, 28, 28)
    for sign in (1.0, -0.6):
        cx, cy = 13.5 + sign * 7 * torch.cos(ang), 13.5 + sign * 7 * torch.sin(ang)
        imgs[:, :, 0] += torch.exp(-((xx[None, None] - cx[..., None, None]) ** 2 + (yy[None, None] - cy[..., None, None]) ** 2) / 8.0)
    return (imgs.clamp(0, 1) - 0.1307) / 0.3081


def _frames(batch):
    """a loader item: the tensor itself (reference loaders, ResidentLoader) or a TensorDataset 1-tuple"""
    return batch[0] if isinstance(batch, (list, tuple)) else batch


def load_data(args):
    if args.task == 'synthetic':
        tr, te = synthetic_sequences(args.Ndata, args.T, args.seed), synthetic_sequences(args.Ntest, args.T, args.seed + 1)
    elif os.path.exists(os.path.join(args.data_root, 'rot_mnist', 'rot-mnist.mat')):
        from .data.wrappers import load_data as load_reference_data
        return load_reference_data(args, plot=False)
    else:
        fn = lambda s: os.path.join(args.data_root, 'rot_mnist_%s.pt' % s)
        if not (os.path.exists(fn('train')) and os.path.exists(fn('test'))):
            raise FileNotFoundError('neither %s nor %s / %s found: the reference dataset is an external download (README.md:19); '
                                    'use --task synthetic to exercise the loop'
                                    % (os.path.join(args.data_root, 'rot_mnist', 'rot-mnist.mat'), fn('train'), fn('test')))
        tr, te = torch.load(fn('train')), torch.load(fn('test'))
    # the shuffle has its own generator: identical order on every rank, whatever else consumes the global RNG
    mk = lambda d, shuffle: torch.utils.data.DataLoader(torch.utils.data.TensorDataset(d), batch_size=args.batch, shuffle=shuffle,
                                                        generator=torch.Generator().manual_seed(args.seed))
    return mk(tr, True), mk(te, False)


def init_distributed():
    """(dist module or None, rank, world, local device index).  WORLD_SIZE > 1 (torchrun): RCCL, one GPU per rank;
    GPODE_DIST_BACKEND=gloo lets several ranks share one card (a rehearsal of the code path, as in bench.py)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world == 1:
        return None, 0, 1, 0
    import torch.distributed as dist
    local = int(os.environ.get('LOCAL_RANK', '0'))
    backend = os.environ.get('GPODE_DIST_BACKEND', 'nccl')
    if backend != 'nccl':
        local %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    else:
        dist.init_process_group(backend)
    return dist, dist.get_rank(), world, local


class BroadcastNoise:
    """Every rank integrates under rank 0's GP draw (the draw is shared by the whole minibatch, odegpvae.py:41-43)."""

    def __init__(self, inner, dist):
        self.inner, self.dist = inner, dist

    def draw(self, *a, **k):
        from .parallel import broadcast_noise
        return broadcast_noise(self.inner.draw(*a, **k), self.dist)


def cache_results(logger, args, ep, build_model):
    """What the reference does when the loss turns NaN (main.py:116-129, called from :205-207) minus the plots: say so, rebuild
    the model, reload the last per-epoch checkpoint in evaluation mode and log its kernel hyper-parameters; returns that model
    (None when no epoch has completed yet -- the reference's torch.load would raise there).  The caller then exits."""
    logger.info('************** Obtained nan Loss at Epoch:{:4d}/{:4d}*************'.format(ep, args.Nepoch))
    logger.info('Laoding previous model for plotting')
    fname = os.path.join(args.save, 'odegpvae_mnist.pth')
    if not os.path.exists(fname):
        logger.info('No checkpoint at {} yet (NaN before the first epoch completed)'.format(fname))
        return None
    model = build_model(args)
    model.to(args.device)
    model.load_state_dict(torch.load(fname, map_location=torch.device(args.device)))
    model.eval()
    kern = model.flow.odefunc.diffeq.kern
    logger.info('Kernel lengthscales {}'.format(kern.lengthscales.data))
    logger.info('Kernel variance {}'.format(kern.variance.data))
    return model


def main(argv=None):
    args = make_parser().parse_args(argv)
    from .model.core.initialization import initialize_and_fix_kernel_parameters
    from .model.create_model import build_model, compute_loss, compute_test_error, backward
    from .model.misc.torch_utils import seed_everything
    from .optim import HipAdam

    if not torch.cuda.is_available():
        raise SystemExit('this build runs on an MI355X (no CPU fallback)')
    dist, rank, world, local = init_distributed()
    args.save = os.path.join(os.path.abspath(os.getcwd()), args.save + datetime.now().strftime('_%d_%m_%Y-%H:%M'), '')
    if rank == 0:
        os.makedirs(os.path.join(args.save, 'plots'), exist_ok=True)
        logger = get_logger(os.path.join(args.save, 'logs'))
    else:                                            # one log / checkpoint per job: rank 0's
        logger = logging.getLogger('gpode_rank%d' % rank)
        logger.addHandler(logging.NullHandler())
        logger.propagate = False
    logger.info('Results stored in {}'.format(args.save))
    seed_everything(args.seed)
    args.device = torch.device('cuda', local) if dist is not None else torch.device('cuda')
    if dist is not None:
        logger.info('Data parallel over {} ranks ({})'.format(world, dist.get_backend()))
    logger.info('Running model on {}'.format(args.device))
    trainset, testset = load_data(args)

    model = build_model(args).to(args.device)
    model = initialize_and_fix_kernel_parameters(model, lengthscale_value=args.lengthscale, variance_value=args.variance, fix=False)
    logger.info(model)
    if args.pretrained:     # main.py:157-170: load the pre-trained VAE, freeze it, encoder / decoder in eval mode
        model.vae.encoder.load_state_dict(torch.load(args.vae_path + '/encoder.pt', map_location=args.device))
        model.vae.decoder.load_state_dict(torch.load(args.vae_path + '/decoder.pt', map_location=args.device))
        for var in model.vae.parameters():
            var.requires_grad = False
        model.vae.encoder.eval()
        model.vae.decoder.eval()
        logger.info('***** Loaded pretrained VAE from {} ********'.format(args.vae_path))
    logger.info('********** Model Built {} ODE **********'.format(args.ode))
    if args.continue_training:
        fname = os.path.join(os.path.abspath(os.getcwd()), args.model_path, 'odegpvae_mnist.pth')
        model.load_state_dict(torch.load(fname, map_location=args.device))
        logger.info('Resume training for model {}'.format(fname))

    meters = {k: RunningAverage(10) for k in ('elbo', 'nll', 'reg_kl', 'inducing_kl')}
    bn_sync = None
    optimizer = HipAdam(model.parameters(), lr=args.lr, bucketed='gather' if dist is not None else False)
    sync = None
    if dist is not None:
        from .parallel import GradAllReduce, shard_batch
        sync = GradAllReduce(optimizer.flat_grads, dist, weight=1.0 / world)
        if args.sync_bn and any(isinstance(m, torch.nn.BatchNorm2d) and m.training for m in model.modules()):
            from . import vae_ops
            from .parallel import BatchNormSync, shard_bounds
            bn_sync = BatchNormSync(dist)
            vae_ops.set_bn_sync(bn_sync)
            logger.info('BatchNorm statistics over the global minibatch ({} ranks)'.format(world))
        gp_layer = model.flow.odefunc.diffeq
        gp_layer.noise_source = BroadcastNoise(gp_layer.noise_source, dist)
        torch.manual_seed(args.seed + 7919 * (rank + 1))     # encoder reparameterisation noise: independent per shard
        torch.cuda.manual_seed(args.seed + 7919 * (rank + 1))
    kern = model.flow.odefunc.diffeq.kern
    graphs = {}     # --hip_graph: one captured step per (L, batch shape), replayed on a static input buffer
    if args.hip_graph:
        from . import ops
        from .graph import GraphedStep, device_generators
        from .model.core.noise import install_device_noise
        install_device_noise(model, args.seed + 1, eps_seed=None if world == 1 else args.seed + 7919 * (rank + 1))
        ops.set_overlap(True)
    elif args.device_noise:
        from .model.core.noise import install_device_noise
        install_device_noise(model, args.seed + 1, eps_seed=None if world == 1 else args.seed + 7919 * (rank + 1))
    if args.gp_side_stream and not args.hip_graph:
        from . import ops
        ops.set_overlap(True)

    # gloo's collectives are host code: a step that holds them (cross-rank BatchNorm) cannot be stream-captured
    capture_ok = not (bn_sync is not None and dist.get_backend() != 'nccl')
    if args.hip_graph and not capture_ok:
        logger.info('cross-rank BatchNorm over {}: steps are launched eagerly (device noise and side stream stay on)'.format(dist.get_backend()))

    def graphed_step(minibatch, L):
        key = (L, tuple(minibatch.shape), bn_sync.shares if bn_sync is not None else None)
        if key not in graphs:
            buf = torch.empty_like(minibatch)

            def step():
                optimizer.zero_grad()
                out = compute_loss(model, buf, L)
                backward(out[0])
                if sync is None:
                    optimizer.step()
                else:                                # the all-reduce and the Adam launch stay between the replays
                    ops.join_side_stream()
                return out
            buf.copy_(minibatch)
            gs = GraphedStep(step, generators=device_generators(model), warmup=1,
                             grad_params=optimizer.params if sync is not None else None)
            graphs[key] = (buf, gs)
            if sync is not None:                     # the gradients the gather / all-reduce below must see are the warm-up's
                gs.bind_warm_grads()
            return gs.warm_out                       # the capture warm-up already took this minibatch's (eager) step
        buf, g = graphs[key]
        buf.copy_(minibatch, non_blocking=True)
        return g()

    from . import ops as gp_ops
    solves_mode = 'auto' if args.backward_solves == 'adaptive' else args.backward_solves
    gp_ops.set_backward_solves(solves_mode)
    logger.info('********** Started Training **********')
    begin = time.time()
    for ep in range(args.Nepoch):
        L = 1 if ep < args.Nepoch // 2 else 5
        for itr, local_batch in enumerate(trainset):
            minibatch = _frames(local_batch).to(args.device)
            if sync is not None and minibatch.shape[0] >= world:     # this rank's shard; its share weights the gradient mean
                n_global = minibatch.shape[0]
                minibatch = shard_batch(minibatch, rank, world)
                sync.weight = minibatch.shape[0] / n_global
                if bn_sync is not None:
                    bn_sync.set_shares([hi - lo for lo, hi in (shard_bounds(n_global, r, world) for r in range(world))])
            elif sync is not None:       # fewer sequences than ranks: every rank takes the whole minibatch
                sync.weight = 1.0 / world
                if bn_sync is not None:
                    bn_sync.set_shares(None)
            if args.hip_graph and capture_ok:
                loss, nlhood, kl_reg, kl_u = graphed_step(minibatch, L)
            else:
                loss, nlhood, kl_reg, kl_u = compute_loss(model, minibatch, L)
            terms = torch.stack([t.detach().reshape(()) for t in (loss, nlhood, kl_reg, kl_u)])
            if sync is not None:                     # the global-batch values: shard means weighted by shard size
                terms = terms * sync.weight
                dist.all_reduce(terms)
            if torch.isnan(terms[0]):
                cache_results(logger, args, ep, build_model)
                if dist is not None:
                    dist.destroy_process_group()
                sys.exit()
            graphed = args.hip_graph and capture_ok
            if not graphed:
                optimizer.zero_grad()
                backward(loss)
            if sync is not None:
                sync.all_reduce_grads()
            if not graphed or sync is not None:
                optimizer.step()
            for k, v in zip(('elbo', 'nll', 'reg_kl', 'inducing_kl'), terms.tolist()):
                meters[k].update(v)
            if itr % args.log_freq == 0:
                logger.info('Iter:{:<2d} | Time {} | elbo {:8.2f}({:8.2f}) | nlhood:{:8.2f}({:8.2f}) | kl_reg:{:<8.2f}({:<8.2f}) | kl_u:{:8.5f}({:8.5f})'.format(
                    itr, timedelta(seconds=time.time() - begin), meters['elbo'].val, meters['elbo'].avg, meters['nll'].val, meters['nll'].avg,
                    meters['reg_kl'].val, meters['reg_kl'].avg, meters['inducing_kl'].val, meters['inducing_kl'].avg))
        if bn_sync is not None:                      # every rank evaluates the SAME test batch: its own statistics are the global ones
            vae_ops.set_bn_sync(None)
        with torch.no_grad():
            for test_batch in testset:               # every rank evaluates (the GP draw is a collective); rank 0 keeps the file
                test_batch = _frames(test_batch)
                test_batch = test_batch.to(args.device)
                Xrec, _, _ = model(test_batch)
                test_mse = compute_test_error(test_batch, Xrec.squeeze(0))
                if rank == 0:
                    torch.save(model.state_dict(), os.path.join(args.save, 'odegpvae_mnist.pth'))
                break
        if bn_sync is not None:
            vae_ops.set_bn_sync(bn_sync)
        if args.backward_solves == 'adaptive':       # once per epoch (the evaluation above has synchronised anyway)
            lo, hi = model.flow.odefunc.diffeq.cache.pivot_range()
            want = 'always' if not (lo > 0) or hi / lo >= 200.0 else 'auto'
            if want != solves_mode:
                solves_mode = want
                gp_ops.set_backward_solves(want)
                graphs.clear()                       # a captured step has its route baked in
                logger.info('Cholesky pivots span {:.1f}x: cache backward route -> {}'.format(hi / lo if lo > 0 else float('inf'), want))
        logger.info('Epoch:{:4d}/{:4d}| tr_elbo:{:8.2f}({:8.2f}) | test_mse:{:5.3f}\n'.format(
            ep, args.Nepoch, meters['elbo'].val, meters['elbo'].avg, test_mse.item()))
    logger.info('********** Optimization completed **********')
    if dist is not None:                             # the ranks must have taken identical steps
        dev = torch.zeros((), device=args.device)
        for p in model.parameters():
            if p.requires_grad:
                ref = p.detach().clone()
                dist.broadcast(ref, src=0)
                dev = torch.maximum(dev, (p.detach() - ref).abs().max())
        dist.all_reduce(dev, op=dist.ReduceOp.MAX)
        logger.info('Largest parameter deviation between ranks: {:.3e}'.format(dev.item()))
    logger.info('Kernel lengthscales {}'.format(kern.lengthscales.data))
    logger.info('Kernel variance {}'.format(kern.variance.data))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
