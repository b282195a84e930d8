"""Data parallelism over the minibatch axis: one process per GPU, torch.distributed over RCCL.

Trajectories are independent given the GP function draw (odegpvae.py:41-43), so every rank runs the same
model on its own shard of the minibatch under the SAME draw (``DeviceNoise`` seeded identically everywhere,
or rank 0's noise broadcast with ``broadcast_noise``).  The only data-path collective is one all-reduce
(mean) of the flat gradient bucket per step: 0.56 MB at cfg1-4.  ``kl_u`` is rank-invariant, the
log-likelihood and KL(z0) terms are per-rank means, so the averaged gradient is exactly the gradient of the
reference loss (create_model.py:72) on the global batch.  BatchNorm is the one operator that couples the samples of a
minibatch (the reference never leaves training mode, vae.py:55,58,113,116,119): ``BatchNormSync`` makes every
BatchNorm layer normalise with the statistics of the GLOBAL minibatch -- an all-gather of 2C+1 floats per layer in the forward
pass and of 2C floats in the backward pass (C <= 64) -- so that an N-rank step is the reference's step on the concatenated batch.
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


class CollectiveStats:
    """Count and bytes of the collectives this package issues (the gradient all-reduce, the BatchNorm statistics all-gathers):
    ``with stats.step(): one_eager_step()`` -> ``stats.per_step`` = {'all_reduce': (count, payload bytes), 'all_gather': ...}.
    xGMI is point-to-point and these messages are tiny, so on N ranks their NUMBER (each a latency-bound ring / tree) is what a
    step pays for, not their bytes."""

    def __init__(self):
        self.on = False
        self.per_step = {}

    def add(self, kind, nbytes):
        if self.on:
            c, b = self.per_step.get(kind, (0, 0))
            self.per_step[kind] = (c + 1, b + int(nbytes))

    class _Step:
        def __init__(self, st):
            self.st = st

        def __enter__(self):
            self.st.per_step, self.st.on = {}, True
            return self.st

        def __exit__(self, *exc):
            self.st.on = False

    def step(self):
        return CollectiveStats._Step(self)

    def summary(self):
        return {k: {'count': c, 'payload_bytes': b} for k, (c, b) in sorted(self.per_step.items())}


stats = CollectiveStats()


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
        stats.add('all_reduce', self.fg.flat.numel() * self.fg.flat.element_size())
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


class BatchNormSync:
    """Cross-rank training-mode BatchNorm (activate with ``vae_ops.set_bn_sync``).  The library splits the layer where the ranks
    exchange numbers (include/gpode.h: gpode_bn_moments / _finalize / _apply, gpode_bn_bwd_sums / _bwd_apply); this class does the
    exchange: an all-gather, so that every rank combines the same values in the same (rank) order.

    ``shares``: sequences per rank of the current global minibatch (default: equal shards).  The data-parallel gradient mean
    weights rank r's loss by shares[r] / sum(shares) (``GradAllReduce.weight``); the backward centring terms need the other
    ranks' sums with that weight relative to this rank's own."""

    def __init__(self, dist, shares=None):
        self.dist = dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        if self.world > 64:
            raise ValueError('BatchNormSync: at most 64 ranks (one wavefront combines the gathered statistics)')
        self._wts = {}
        self.set_shares(shares)

    def set_shares(self, shares=None):
        self.shares = tuple(float(v) for v in shares) if shares is not None else (1.0,) * self.world
        if len(self.shares) != self.world or min(self.shares) <= 0:
            raise ValueError('BatchNormSync.set_shares: one positive share per rank')

    def gather(self, t):
        """t (K,) on every rank -> (world, K), row r = rank r's t."""
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        stats.add('all_gather', t.numel() * t.element_size())
        self.dist.all_gather(list(out.unbind(0)), t)
        return out

    def gather_many(self, tensors):
        """Several INDEPENDENT statistics vectors in one collective (e.g. the same-depth BatchNorm layers of the position and the
        velocity encoder of a second-order model, vae.py:14-19): packed into one payload, all-gathered once, handed back as the
        (world, K_i) tensors separate ``gather`` calls would return -- the same values, bit for bit, in the same rank order."""
        flat = torch.cat([t.reshape(-1) for t in tensors])
        got = self.gather(flat)
        out, o = [], 0
        for t in tensors:
            out.append(got[:, o:o + t.numel()].reshape((self.world,) + tuple(t.shape)).contiguous())
            o += t.numel()
        return out

    def weights(self, device):
        """(world,) device tensor shares[r] / shares[rank] (cached per share pattern)."""
        key = (self.shares, str(device))
        if key not in self._wts:
            self._wts[key] = torch.tensor([v / self.shares[self.rank] for v in self.shares], dtype=torch.float32, device=device)
        return self._wts[key]

    def count_all(self, local_count):
        """global element count of a layer whose local element count is ``local_count`` (host arithmetic, no communication)."""
        return float(local_count) * (sum(self.shares) / self.shares[self.rank])
