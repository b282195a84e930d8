"""VAE pre-training loop -- the training part of experiments/main_vae.py (argument surface :18-50, loop :56-129): encoder,
reparameterisation, decoder, Bernoulli log-likelihood (``pretrain=True``: images, no time axis), KL to N(0, I), Adam; writes
``<output_path>_<date>/MNIST-VAE/{encoder,decoder}.pt``, the files ``main.py --pretrained True --vae_path ...`` loads.
Every arithmetic step runs in the HIP library.  The reference builds its images from torchvision's MNIST
(data/mnist.py:167-172, a download); ``--synthetic True`` substitutes seeded random frames of the same shape, a saved
``rotating_mnist_train_3_<n_angle>_angles.npy`` under ``--save`` is used as is."""
import argparse
import os
import time
from datetime import datetime, timedelta

import numpy as np
import torch

# the reference's flags, in its order (main_vae.py:18-50)
FLAGS = [
    ('digit', int, 3, 'For which digit to create the training data'), ('n_angle', int, 16, 'Data set time steps of a full rotation'),
    ('n_train', int, 180, 'number of training sequences'), ('n_test', int, 121, 'number of test sequences'),
    ('batch', int, 64, 'batch size'), ('latent_dim', int, 6, 'latent dimensionality'), ('device', str, 'cuda:0', 'device'),
    ('lr', float, 0.001, 'Learning rate for model training'), ('seed', int, 121, 'Global seed for the training run'),
    ('vae_epochs', int, 300, 'Number of epochs'), ('output_path', str, 'results/vae', 'Directory name for saving all the model outputs'),
    ('save', str, 'data/moving_mnist', 'Directory of the rotating-MNIST arrays'), ('log_freq', int, 20, 'Logging frequency while training'),
]
EXT_FLAGS = [('synthetic', eval, False, 'train on seeded random frames instead of the (downloaded) rotating MNIST arrays')]


def make_parser():
    p = argparse.ArgumentParser('Learning Latent Encoding with VAE')
    for name, typ, default, hlp in FLAGS + EXT_FLAGS:
        p.add_argument('--' + name, type=typ, default=default, help=hlp)
    return p


def load_images(args):
    """(n_train * n_angle, 1, 28, 28) float32 in [0, 1] (data/mnist.py:25-88 flattens the rotation sequences to frames)."""
    if args.synthetic:
        g = torch.Generator().manual_seed(args.seed)
        return torch.rand(args.n_train * args.n_angle, 1, 28, 28, generator=g)
    fn, fn_test = (os.path.join(args.save, 'rotating_mnist_%s_3_%d_angles.npy' % (s, args.n_angle)) for s in ('train', 'test'))
    if not os.path.exists(fn):      # main_vae.py:155-165: build the arrays once and keep them next to the digits
        from .data.mnist import create_rotating_dataset
        try:
            train, test = create_rotating_dataset(args.save, digit=args.digit, train_n=args.n_train, test_n=args.n_test, n_angles=args.n_angle)
        except FileNotFoundError as e:
            raise FileNotFoundError('%s not found and it cannot be created (%s); use --synthetic True to exercise the loop' % (fn, e))
        np.save(fn, train)
        np.save(fn_test, test)
    return torch.tensor(np.load(fn), dtype=torch.float32).reshape(-1, 1, 28, 28)


def vae_train(args, images, epochs, output_model_path, log=print):
    from . import vae_ops as V
    from .main import RunningAverage
    from .model.core.vae import VAE
    from .optim import HipAdam
    os.makedirs(output_model_path, exist_ok=True)
    vae = VAE(device=args.device, latent_dim=args.latent_dim).to(args.device)
    vae.print_summary()
    opt = HipAdam(list(vae.encoder.parameters()) + list(vae.decoder.parameters()), lr=args.lr, bucketed=False)
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(images), batch_size=args.batch, shuffle=True)
    meters = {k: RunningAverage(10) for k in ('elbo', 'nll', 'reg_kl')}
    begin = time.time()
    for ep in range(epochs):
        vae.encoder.train()
        vae.decoder.train()
        for itr, (x,) in enumerate(loader):
            opt.zero_grad()
            x = x.to(args.device)
            mu, logvar = vae.encoder(x)
            z = vae.encoder.sample(mu, logvar)
            kl_reg = vae.encoder.kl_rows(mu, logvar).mean(0)                       # kl(q, prior).sum(-1).mean(0)
            y = vae.decoder(z)
            lhood = V.bernoulli_loglik_rowsum(x, y, x.shape[0]).mean(0)            # log_prob(x, y, pretrain=True).sum([1,2,3]).mean(0)
            loss = kl_reg - lhood
            loss.backward()
            opt.step()
            for k, v in zip(('elbo', 'nll', 'reg_kl'), (loss, -lhood, kl_reg)):
                meters[k].update(v.item())
            if itr % args.log_freq == 0:
                log('Iter:{:<2d} | Time {} | elbo {:8.2f}({:8.2f}) | nlhood:{:8.2f}({:8.2f}) | kl_reg:{:<8.2f}({:<8.2f})'.format(
                    itr, timedelta(seconds=time.time() - begin), meters['elbo'].val, meters['elbo'].avg, meters['nll'].val,
                    meters['nll'].avg, meters['reg_kl'].val, meters['reg_kl'].avg))
        log('Epoch:{:4d}/{:4d}| tr_elbo:{:8.2f}({:8.2f}))\n'.format(ep, epochs, meters['elbo'].val, meters['elbo'].avg))
    vae.save(os.path.join(output_model_path, 'encoder.pt'), os.path.join(output_model_path, 'decoder.pt'))
    return vae, meters


def main(argv=None):
    from .main import get_logger
    from .model.misc.torch_utils import seed_everything
    args = make_parser().parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit('this build runs on an MI355X (no CPU fallback)')
    args.output_path = os.path.join(os.path.abspath(os.getcwd()), args.output_path + datetime.now().strftime('_%d_%m_%Y-%H:%M'), '')
    os.makedirs(os.path.join(args.output_path, 'plots'), exist_ok=True)
    logger = get_logger(os.path.join(args.output_path, 'logs'))
    logger.info('Results stored in {}'.format(args.output_path))
    seed_everything(args.seed)
    logger.info('Running model on {}'.format(args.device))
    logger.info('Model parameters:  num epochs {} | lr {} | latent_dim {} | n_angles {}'.format(
        args.vae_epochs, args.lr, args.latent_dim, args.n_angle))
    vae, meters = vae_train(args, load_images(args), args.vae_epochs, os.path.join(args.output_path, 'MNIST-VAE'), log=logger.info)
    return os.path.join(args.output_path, 'MNIST-VAE')


if __name__ == '__main__':
    main()
