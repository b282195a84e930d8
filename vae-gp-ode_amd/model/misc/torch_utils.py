"""Host utilities the model mirror needs from experiments/model/misc/torch_utils.py: the two reshaping modules that sit in
``Encoder.cnn`` / ``Decoder.decnn`` (:16-22; parameter-free, they only have to exist so that the Sequential indices -- and with
them the state_dict keys ``cnn.0 ... decnn.10`` -- come out as in the reference) and ``seed_everything`` (:64-73)."""
import os
import random

import numpy as np
import torch


class Flatten(torch.nn.Module):
    """(N, C, H, W) -> (N, C*H*W)"""

    def forward(self, x):
        return x.flatten(1)


class UnFlatten(torch.nn.Module):
    """(N, C*w*w) -> (N, C, w, w)"""

    def __init__(self, w):
        super().__init__()
        self.w = w

    def forward(self, x):
        return x.unflatten(1, (-1, self.w, self.w))


def seed_everything(seed):
    """Seed python, numpy and torch (host and device generators) and pin the hash seed, as main.py:142 expects."""
    os.environ['PYTHONHASHSEED'] = str(seed)
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
