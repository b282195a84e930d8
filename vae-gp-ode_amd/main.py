"""Training entry point with the argument surface of the reference's experiments/main.py (flags, types and
defaults of main.py:23-114; loop contract of main.py:199-247): L = 1 for the first half of the epochs then 5,
NaN guard, per-epoch evaluation on the first test batch, ``odegpvae_mnist.pth`` checkpoint every epoch,
``--continue_training`` / ``--pretrained`` wiring, same log lines.  Every arithmetic step runs in the HIP
kernels behind ``model/``; the optimizer is the one-launch HIP Adam.

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
]
CHOICES = {'kernel': KERNELS, 'solver': SOLVERS}  # default 'euler' is accepted only as a default, as in the reference


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
    mk = lambda d, shuffle: torch.utils.data.DataLoader(torch.utils.data.TensorDataset(d), batch_size=args.batch, shuffle=shuffle)
    return mk(tr, True), mk(te, False)


def main(argv=None):
    args = make_parser().parse_args(argv)
    from .model.core.initialization import initialize_and_fix_kernel_parameters
    from .model.create_model import build_model, compute_loss, compute_test_error
    from .model.misc.torch_utils import seed_everything
    from .optim import HipAdam

    args.save = os.path.join(os.path.abspath(os.getcwd()), args.save + datetime.now().strftime('_%d_%m_%Y-%H:%M'), '')
    os.makedirs(os.path.join(args.save, 'plots'), exist_ok=True)
    logger = get_logger(os.path.join(args.save, 'logs'))
    logger.info('Results stored in {}'.format(args.save))
    seed_everything(args.seed)
    if not torch.cuda.is_available():
        raise SystemExit('this build runs on an MI355X (no CPU fallback)')
    args.device = torch.device('cuda')
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
    optimizer = HipAdam(model.parameters(), lr=args.lr)
    kern = model.flow.odefunc.diffeq.kern
    graphs = {}     # --hip_graph: one captured step per (L, batch shape), replayed on a static input buffer
    if args.hip_graph:
        from . import ops
        from .graph import GraphedStep, device_generators
        from .model.core.noise import DeviceNoise
        model.flow.odefunc.diffeq.noise_source = DeviceNoise(args.seed + 1)
        ops.set_overlap(True)

    def graphed_step(minibatch, L):
        key = (L, tuple(minibatch.shape))
        if key not in graphs:
            buf = torch.empty_like(minibatch)

            def step():
                optimizer.zero_grad()
                out = compute_loss(model, buf, L)
                out[0].backward()
                optimizer.step()
                return out
            buf.copy_(minibatch)
            gp = model.flow.odefunc.diffeq
            gp.noise_source.draw(gp.kernel_n, gp.D_in, gp.D_out, gp.M, gp.S, minibatch.device)   # creates the generator
            graphs[key] = (buf, GraphedStep(step, generators=device_generators(model), warmup=1))
            return graphs[key][1].warm_out           # the capture warm-up already took this minibatch's (eager) step
        buf, g = graphs[key]
        buf.copy_(minibatch, non_blocking=True)
        return g()

    logger.info('********** Started Training **********')
    begin = time.time()
    for ep in range(args.Nepoch):
        L = 1 if ep < args.Nepoch // 2 else 5
        for itr, local_batch in enumerate(trainset):
            minibatch = _frames(local_batch).to(args.device)
            if args.hip_graph:
                loss, nlhood, kl_reg, kl_u = graphed_step(minibatch, L)
            else:
                loss, nlhood, kl_reg, kl_u = compute_loss(model, minibatch, L)
            if torch.isnan(loss):
                logger.info('************** Obtained nan Loss at Epoch:{:4d}/{:4d}*************'.format(ep, args.Nepoch))
                sys.exit()
            if not args.hip_graph:
                optimizer.zero_grad()
                loss.backward()
                optimizer.step()
            for k, v in zip(('elbo', 'nll', 'reg_kl', 'inducing_kl'), (loss, nlhood, kl_reg, kl_u)):
                meters[k].update(v.item())
            if itr % args.log_freq == 0:
                logger.info('Iter:{:<2d} | Time {} | elbo {:8.2f}({:8.2f}) | nlhood:{:8.2f}({:8.2f}) | kl_reg:{:<8.2f}({:<8.2f}) | kl_u:{:8.5f}({:8.5f})'.format(
                    itr, timedelta(seconds=time.time() - begin), meters['elbo'].val, meters['elbo'].avg, meters['nll'].val, meters['nll'].avg,
                    meters['reg_kl'].val, meters['reg_kl'].avg, meters['inducing_kl'].val, meters['inducing_kl'].avg))
        with torch.no_grad():
            for test_batch in testset:
                test_batch = _frames(test_batch)
                test_batch = test_batch.to(args.device)
                Xrec, _, _ = model(test_batch)
                test_mse = compute_test_error(test_batch, Xrec.squeeze(0))
                torch.save(model.state_dict(), os.path.join(args.save, 'odegpvae_mnist.pth'))
                break
        logger.info('Epoch:{:4d}/{:4d}| tr_elbo:{:8.2f}({:8.2f}) | test_mse:{:5.3f}\n'.format(
            ep, args.Nepoch, meters['elbo'].val, meters['elbo'].avg, test_mse.item()))
    logger.info('********** Optimization completed **********')
    logger.info('Kernel lengthscales {}'.format(kern.lengthscales.data))
    logger.info('Kernel variance {}'.format(kern.variance.data))


if __name__ == '__main__':
    main()
