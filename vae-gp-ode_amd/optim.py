"""Adam in one HIP launch over the whole parameter list (csrc/vae_conv.hip:k_adam_multi); same update rule
and defaults as ``torch.optim.Adam(params, lr)`` at experiments/main.py:194."""
import ctypes

import torch

from . import _lib
from . import ops
from .ops import _stream
from .parallel import FlatGrads


class HipAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, bucketed=True):
        """bucketed=True: every ``.grad`` is a persistent view into one flat buffer (what the data-parallel all-reduce
        needs); autograd then ADDS each produced gradient into it (one small kernel per parameter).  bucketed=False
        (single GPU): ``zero_grad`` drops the gradients, autograd hands over the tensors the backward kernels wrote,
        and the update reads them through a pointer table -- no zero fill, no per-parameter add.
        bucketed='gather' (data parallelism): gradients are handed over as with False; ``gather_grads()`` copies them into
        the flat bucket in ONE launch (called by ``GradAllReduce`` before the collective) and the update reads the bucket."""
        self.params = [p for p in params if p.requires_grad]
        if not self.params or not all(p.is_cuda and p.dtype == torch.float32 for p in self.params):
            raise _lib.GpodeError('HipAdam needs float32 CUDA/HIP parameters')
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        dev = self.params[0].device
        # device-resident copy of the step count: incremented by the update kernel's launch sequence, so a captured
        # HIP graph of the whole training step (see graph.py) replays with the right bias correction
        self.step_dev = torch.zeros(2, dtype=torch.int32, device=dev)     # {updates done, ticket}: advanced by the update kernel
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        # persistent gradient storage: views into one flat bucket (all-reduced in place under data parallelism)
        self.gather = bucketed == 'gather'
        self.bucketed = bool(bucketed) and not self.gather
        self.flat_grads = FlatGrads(self.params, views=not self.gather) if bucketed else None
        self._gathered = False
        if self.gather:
            self.flat_grads.gather = self.gather_grads

        # Handed-over gradients (False / 'gather') are first READ by this optimiser (or the bucket gather), after the backward
        # pass has ended: the backward kernels' final reductions of workgroup partials may then all run in ONE launch at the end of
        # the pass (vae_ops.set_deferred_reductions).  With persistent views autograd adds every gradient as it arrives -- too early.
        self.allows_deferred_reductions = not self.bucketed
        from . import vae_ops
        vae_ops.set_deferred_reductions(self.allows_deferred_reductions)   # the optimiser in charge decides (process-wide switch)
        self._zero = {}
        offs, tot = [], 0
        for p in self.params:
            offs.append(tot)
            tot += p.numel()
        self.total = tot
        tab = lambda ts: torch.tensor([t.data_ptr() for t in ts], dtype=torch.int64, device=dev)
        self._p, self._m, self._v = tab(self.params), tab(self.exp_avg), tab(self.exp_avg_sq)
        # gradient pointer tables: one small device tensor per distinct set of gradient addresses, uploaded ONCE (the first time
        # the set is seen) and kept.  Graph replays hand over the same tensors every time and the eager allocator cycles through
        # a few sets, so the steady state uploads nothing.  (A single table refreshed through one pinned staging buffer is a race:
        # the host may overwrite the staging buffer for step i+1 before the device has run step i's asynchronous copy.)
        self._tables = {}
        self._g = None
        # pinned sources for tables uploaded INSIDE a graph capture (allocating pinned memory is not a capturable call): one per
        # captured graph that holds the update
        self._pinned = [torch.zeros(len(self.params), dtype=torch.int64).pin_memory() for _ in range(8)]
        self._static_tabs = [torch.zeros(len(self.params), dtype=torch.int64, device=dev) for _ in range(8)]   # see _refresh_grad_table
        self._offs = torch.tensor(offs, dtype=torch.int64, device=dev)
        if self.gather:                              # the update reads the (all-reduced) bucket: a static table of pointers into it
            base = self.flat_grads.flat.data_ptr()
            self._gflat = torch.tensor([base + 4 * o for o in offs], dtype=torch.int64, device=dev)

    def zero_grad(self):
        if not self.bucketed:
            for p in self.params:
                p.grad = None
            return
        fg = self.flat_grads
        for p, o in zip(fg.params, fg.offsets):
            if p.grad is None or p.grad.data_ptr() != fg.flat.data_ptr() + 4 * o:
                p.grad = fg.flat[o:o + p.numel()].view_as(p)
        fg.zero()

    def _refresh_grad_table(self):
        cur = []
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None:                            # parameter not reached by this backward
                g = self._zero.get(i)
                if g is None:
                    g = self._zero[i] = torch.zeros_like(p)
            elif not g.is_contiguous() or g.dtype != torch.float32:
                raise _lib.GpodeError('HipAdam: gradients must be contiguous float32')
            cur.append(g.data_ptr())
        key = tuple(cur)
        tab = self._tables.get(key)
        if tab is None:
            if len(self._tables) >= 256:             # an allocator that never repeats itself: do not grow without bound
                self._tables.clear()
            if torch.cuda.is_current_stream_capturing():
                # inside a capture the upload must be a node of the graph: pinned source kept alive with the table
                if not self._pinned or not self._static_tabs:
                    raise _lib.GpodeError('HipAdam: more than 8 captured graphs hold the update; create the optimiser with more staging buffers')
                from . import graph
                if graph.capturing_step():
                    # GraphedStep: the table's content is a constant of the captured graph -- it is uploaded ONCE, right after the
                    # capture ends and before the first replay, instead of by a copy node that every replay would repeat.  The
                    # table must then live OUTSIDE the graph's memory pool: a block allocated during the capture may have had an
                    # earlier life inside the same step, and the kernel that wrote it then overwrites the table at every replay.
                    tab = self._static_tabs.pop()
                    vals = torch.tensor(cur, dtype=torch.int64)
                    graph.after_capture(lambda t=tab, v=vals: t.copy_(v))
                    self._tables[key] = (tab, vals)
                else:
                    tab = torch.empty(len(cur), dtype=torch.int64, device=self._p.device)
                    host = self._pinned.pop()
                    host.copy_(torch.tensor(cur, dtype=torch.int64))
                    tab.copy_(host, non_blocking=True)
                    self._tables[key] = (tab, host)
            else:
                tab = torch.tensor(cur, dtype=torch.int64, device=self._p.device)   # synchronous upload, once per set
                self._tables[key] = (tab, None)
            tab = self._tables[key]
        self._g = tab[0]

    def gather_grads(self):
        """bucketed='gather': fill the flat bucket from the gradients autograd produced (one launch)."""
        ops.join_side_stream()
        self._refresh_grad_table()
        vp = lambda t: ctypes.c_void_p(t.data_ptr())
        _lib.call('gpode_gather_multi', vp(self._g), vp(self._offs), len(self.params), self.total, vp(self.flat_grads.flat), _stream())
        self._gathered = True

    def step(self):
        ops.join_side_stream()                       # overlap mode: deferred GP parameter gradients land in .grad here
        vp = lambda t: ctypes.c_void_p(t.data_ptr())
        if self.gather:
            if not self._gathered:                   # no collective in between (single rank): the bucket is still to be filled
                self.gather_grads()
            self._gathered = False
            gtab = self._gflat
        else:
            self._refresh_grad_table()
            gtab = self._g
        self.step_count += 1
        _lib.call('gpode_adam_multi', vp(self._p), vp(gtab), vp(self._m), vp(self._v), vp(self._offs), len(self.params),
                  self.total, ctypes.c_float(self.lr), ctypes.c_float(self.betas[0]), ctypes.c_float(self.betas[1]),
                  ctypes.c_float(self.eps), self.step_count, vp(self.step_dev), _stream())
