"""Rotating-MNIST loaders with the names and outputs of experiments/data/mnist.py.

* ``load_mnist_data(args, plot)`` (mnist.py:25-88): ``rot_mnist/rot-mnist.mat`` (keys ``X`` (n, 16, 784) in [0, 1], ``Y`` digit
  labels) -> the digit-``value`` sequences; the first ``Ndata`` train, the ``Ntest`` after the ``Ntest``-long validation
  block test (mnist.py:33-35,46-53: N = 360, valid = 40, test = 40 -- the values the flags default to).  The reference hard-codes
  those numbers and the relative path ``data/``; here they are read from ``args`` (same defaults).  Items are z-normalised
  (16, 1, 28, 28) tensors (``Dataset``).  With ``args.device`` on the GPU the loaders are ``ResidentLoader``s (set lives in
  HBM); ``args.resident = False`` gives the reference's host ``DataLoader``s.
* ``load_mat_mnist_data`` (mnist.py:91-129) and ``load_rotating_mnist_data`` (mnist.py:131-147): frame-level loaders with the
  rotation index as label, used by the VAE pre-training.
* ``rot_start`` (mnist.py:14-22), ``rotate_img`` (mnist.py:150-161), ``create_rotating_dataset`` (mnist.py:163-193).
  The last needs the MNIST digits: torchvision (a download in the reference) is not in this image, so the raw arrays can be
  passed in (``images=``, ``labels=``, ...) or read from ``<data_path>/mnist.npz``.
"""
import os

import numpy as np
import torch
from torch.utils import data

from .utils import Dataset, Dataset_labels, ResidentLoader, normalise


def rot_start(Xtr, T, N):
    """Random starting angle s per sequence (N, T, 1, 28, 28): frames s..T-1, then frames 1..s (mnist.py:14-22; its
    ``torch.flip(..., dims=(1,))`` acts on the channel axis of the slice, a no-op for one channel, and is kept as is).
    The N starts come from numpy's global generator like the reference's."""
    starts = np.random.randint(0, T, N)
    out = []
    for n in range(N):
        s = int(starts[n])
        out.append(torch.cat((Xtr[n, s:], torch.flip(Xtr[n, 1:s + 1], dims=(1,))), dim=0))
    return torch.stack(out, 0)


def _digit_sequences(args):
    import scipy.io as sio
    path = os.path.join(getattr(args, 'data_root', 'data'), 'rot_mnist', 'rot-mnist.mat')
    if not os.path.exists(path):
        raise FileNotFoundError('%s not found: the rotating-MNIST file is an external download (README.md:19)' % path)
    mat = sio.loadmat(path)
    X = np.squeeze(mat['X'])
    if getattr(args, 'mask', True):
        X = X[np.squeeze(mat['Y']) == getattr(args, 'value', 3)]
    return X


def _plot_grid(frames, rows, cols, fname, size):
    import matplotlib
    matplotlib.use('agg')
    import matplotlib.pyplot as plt
    fig, axs = plt.subplots(rows, cols, figsize=size, squeeze=False)
    for ax, img in zip(axs.flat, frames):
        ax.imshow(np.asarray(img).reshape(28, 28), cmap='gray')
        ax.axis('off')
    os.makedirs(os.path.dirname(fname), exist_ok=True)
    fig.savefig(fname)
    plt.close(fig)


def _sequence_loader(seqs, args, shuffle=True):
    device = torch.device(getattr(args, 'device', 'cpu'))
    if device.type == 'cuda' and getattr(args, 'resident', True):
        items = normalise(torch.as_tensor(np.ascontiguousarray(seqs), dtype=torch.float32).reshape(len(seqs), -1, 1, 28, 28))
        return ResidentLoader(items, args.batch, shuffle=shuffle, device=device, seed=getattr(args, 'seed', 0))
    return data.DataLoader(Dataset(seqs), batch_size=args.batch, shuffle=shuffle, num_workers=0)


def load_mnist_data(args, plot=True):
    X = _digit_sequences(args)
    N, held = getattr(args, 'Ndata', 360), getattr(args, 'Ntest', 40)
    if X.shape[0] < N + 2 * held:
        raise ValueError('rot-mnist.mat holds %d sequences of the digit; Ndata + 2 Ntest = %d needed' % (X.shape[0], N + 2 * held))
    trainset = _sequence_loader(X[:N], args)
    testset = _sequence_loader(X[N + held:N + 2 * held], args)          # X[N:N+held] is the (unused) validation block
    if plot:
        first = next(iter(trainset)).cpu()
        _plot_grid(first[:6].reshape(-1, 784), min(6, first.shape[0]), first.shape[1], os.path.join(args.save, 'plots/data.png'), (20, 8))
    return trainset, testset


def load_mat_mnist_data(args, plot=True):
    X = _digit_sequences(args)
    N, T = args.Ndata, args.T
    frames = lambda seqs: torch.tensor(seqs, dtype=torch.float32).reshape(-1, 1, 28, 28)
    Xtr, Xte = frames(X[:N]), frames(X[N:N + args.Ntest])
    angle = np.arange(T, dtype=np.uint8)[None]
    labels = lambda x: np.repeat(angle, x.shape[0] // T, axis=0).reshape(-1, 1)
    mk = lambda x: data.DataLoader(Dataset_labels(x, labels(x)), batch_size=args.batch, shuffle=True, num_workers=0)
    trainset, testset = mk(Xtr), mk(Xte)
    if plot:
        x, _ = next(iter(trainset))
        _plot_grid(x[:16], 4, 4, os.path.join(args.save, 'plots/data.png'), (8, 8))
    return trainset, testset


def load_rotating_mnist_data(data_path, args, plot=True):
    x_true = np.load(data_path).reshape(-1, 1, 28, 28)
    angle = np.arange(args.n_angle, dtype=np.uint8)[None]
    labels = np.repeat(angle, x_true.shape[0] // args.n_angle, axis=0).reshape(-1, 1)
    loader = data.DataLoader(Dataset_labels(x_true, labels), batch_size=args.batch, shuffle=True)
    if plot:
        _plot_grid(x_true[:args.n_angle], 1, args.n_angle, os.path.join(args.save, 'sample-dataset.png'), (120, 5))
    return loader


def rotate_img(img, angles):
    """(n, 28, 28) -> (n, 1 + len(angles), 28, 28): the image followed by its rotations (degrees; scipy.ndimage.rotate's default
    cubic spline, reshape=False), mnist.py:150-161."""
    from scipy.ndimage import rotate
    base = np.asarray(img)
    views = [base] + [rotate(base, a, axes=(1, 2), reshape=False) for a in angles]
    return np.stack([v.reshape(-1, 28, 28) for v in views], axis=1)


def create_rotating_dataset(data_path, digit=3, train_n=100, test_n=10, n_angles=64, images=None, labels=None,
                            test_images=None, test_labels=None):
    """``train_n`` / ``test_n`` randomly chosen images of ``digit`` (from the MNIST train / test split) rotated through
    ``n_angles`` equally spaced angles of a full turn, scaled to [0, 1] float32: (n, n_angles, 28, 28) each
    (mnist.py:163-193).  Indices come from numpy's global generator, train first, as in the reference."""
    if images is None:
        npz = os.path.join(data_path, 'mnist.npz')
        if not os.path.exists(npz):
            raise FileNotFoundError('%s not found: torchvision (the reference\'s MNIST download) is not available here; pass '
                                    'images= / labels= (+ test_images= / test_labels=) or provide the archive with keys '
                                    'x_train, y_train, x_test, y_test (uint8 (n,28,28) / labels)' % npz)
        with np.load(npz) as z:
            images, labels, test_images, test_labels = z['x_train'], z['y_train'], z['x_test'], z['y_test']
    if test_images is None:
        test_images, test_labels = images, labels
    angles = np.rad2deg(np.linspace(0, 2 * np.pi, n_angles)[1:])
    sets = []
    for n, (x, y) in ((train_n, (images, labels)), (test_n, (test_images, test_labels))):
        pool = np.asarray(x)[np.asarray(y) == digit]
        pick = pool[np.random.randint(0, pool.shape[0], n)]
        sets.append((rotate_img(pick, angles) / 255).astype(np.float32))
    return sets[0], sets[1]
