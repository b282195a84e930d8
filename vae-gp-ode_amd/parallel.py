"""Data parallelism over the minibatch axis: one process per GPU, torch.distributed over RCCL.

Trajectories are independent given the GP function draw (odegpvae.py:41-43), so every rank runs the same
model on its own shard of the minibatch under the SAME draw (``DeviceNoise`` seeded identically everywhere,
or rank 0's noise broadcast with ``broadcast_noise``).  The only data-path collective is one all-reduce
(mean) of the flat gradient bucket per step: 0.56 MB at cfg1-4.  ``kl_u`` is rank-invariant, the
log-likelihood and KL(z0) terms are per-rank means, so the averaged gradient is exactly the gradient of the
reference loss (create_model.py:72) on the global batch -- except for BatchNorm, which normalises with
per-rank batch statistics here (the reference's train-mode BatchNorm couples the whole batch).
"""
import torch


def shard_bounds(n, rank, world):
    """[lo, hi) of rank's shard of n items; shards differ by at most one item."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(X, rank, world):
    lo, hi = shard_bounds(X.shape[0], rank, world)
    return X[lo:hi]


class FlatGrads:
    """One contiguous gradient buffer; every parameter's ``.grad`` is a view into it, so the bucket is
    all-reduced (and handed to the fused Adam) without gather/scatter copies."""

    def __init__(self, params, views=True):
        """views=False: only the bucket is allocated; ``gather`` (set by the optimiser) fills it from the produced gradients."""
        self.params = [p for p in params if p.requires_grad]
        self.gather = None
        self.offsets, tot = [], 0
        for p in self.params:
            self.offsets.append(tot)
            tot += p.numel()
        self.total = tot
        p0 = self.params[0]
        self.flat = torch.zeros(tot, dtype=p0.dtype, device=p0.device)
        if views:
            for p, o in zip(self.params, self.offsets):
                p.grad = self.flat[o:o + p.numel()].view_as(p)

    def zero(self):
        self.flat.zero_()


class GradAllReduce:
    def __init__(self, flat_grads, dist, weight=None):
        """weight: this rank's share of the global batch (n_local / n_global) for uneven shards; default 1/world."""
        self.fg, self.dist = flat_grads, dist
        self.world = dist.get_world_size()
        self.weight = weight
        self.avg = dist.get_backend() == 'nccl' and weight is None

    def all_reduce_grads(self):
        if self.fg.flat.is_cuda:
            from . import ops
            ops.join_side_stream()                   # overlap mode: the GP parameter gradients must be in the bucket first
        if self.fg.gather is not None:
            self.fg.gather()                         # bucket filled in one launch from the tensors autograd handed over
        if self.avg:
            self.dist.all_reduce(self.fg.flat, op=self.dist.ReduceOp.AVG)
        else:
            self.fg.flat.mul_(self.weight if self.weight is not None else 1.0 / self.world)
            self.dist.all_reduce(self.fg.flat, op=self.dist.ReduceOp.SUM)


def broadcast_noise(noise, dist, src=0):
    """Make every rank integrate under rank `src`'s GP draw."""
    for k in sorted(noise):
        dist.broadcast(noise[k], src=src)
    return noise
