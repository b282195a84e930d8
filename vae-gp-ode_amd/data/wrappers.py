"""``load_data(args, plot=False)`` dispatch of experiments/data/wrappers.py:3-5."""
from .mnist import load_mnist_data


def load_data(args, plot=False):
    if args.task != 'mnist':
        raise ValueError("task '%s': only 'mnist' has a loader (wrappers.py:4)" % args.task)
    return load_mnist_data(args, plot=plot)
