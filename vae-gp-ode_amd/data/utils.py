"""Dataset wrappers of experiments/data/utils.py and the device-resident loader this build prefers.

``Dataset`` (utils.py:5-15) normalises every item with the MNIST statistics (0.1307, 0.3081) and reshapes it to
(16, 1, 28, 28); ``Dataset_labels`` (utils.py:17-28) pairs frames with their time index.  Both are host-side and feed
``torch.utils.data.DataLoader`` exactly like the reference's.

``ResidentLoader``: rotating MNIST is 360 x 16 x 784 floats = 18 MB -- nothing on a 288 GB device.  The normalised set is
uploaded once; an epoch is a device-side permutation and every minibatch one ``index_select`` on the current stream, so the
training step (a replayed HIP graph reading a static input buffer) never waits for a host copy.
"""
import torch
from torch.utils import data

MNIST_MEAN, MNIST_STD = 0.1307, 0.3081


def normalise(frames):
    return (frames - MNIST_MEAN) / MNIST_STD


class Dataset(data.Dataset):
    """Sequences (N, 16, 784) in [0, 1] -> items (16, 1, 28, 28), z-normalised (utils.py:5-15)."""

    def __init__(self, Xtr):
        self.Xtr = Xtr
        self.mean, self.std = MNIST_MEAN, MNIST_STD

    def __len__(self):
        return len(self.Xtr)

    def __getitem__(self, idx):
        item = torch.as_tensor(self.Xtr[idx], dtype=torch.float32).reshape(16, 1, 28, 28)
        return (item - self.mean) / self.std


class Dataset_labels(data.Dataset):
    """(frame, label) pairs; the label array is flattened (utils.py:17-28)."""

    def __init__(self, x, y):
        self.x, self.y = x, y.reshape(-1)

    def __len__(self):
        return self.y.shape[0]

    def __getitem__(self, index):
        return self.x[index], self.y[index]


class ResidentLoader:
    """Iterates minibatches of a tensor that stays on ``device``.  ``len()`` and the ragged last batch follow DataLoader
    (drop_last=False); ``shuffle`` draws one permutation per epoch from a device generator seeded with ``seed``."""

    def __init__(self, items, batch_size, shuffle=True, device='cuda', seed=0):
        self.items = items.to(device).contiguous()
        self.batch_size, self.shuffle = int(batch_size), shuffle
        self.gen = torch.Generator(device=self.items.device).manual_seed(seed)

    @property
    def dataset(self):
        return self.items

    def __len__(self):
        return (self.items.shape[0] + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = self.items.shape[0]
        order = torch.randperm(n, device=self.items.device, generator=self.gen) if self.shuffle else None
        for lo in range(0, n, self.batch_size):
            yield self.items[lo:lo + self.batch_size] if order is None else self.items.index_select(0, order[lo:lo + self.batch_size])
