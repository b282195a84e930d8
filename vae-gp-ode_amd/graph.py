"""Replay a whole training step as one HIP graph.

A step of the ELBO loop (experiments/main.py:198-212: zero_grad, compute_loss, backward, optimizer step) is ~250
kernel launches, half of them a few microseconds long; launched one by one the GPU idles between them.  Every
op of this package launches on torch's current stream and allocates through torch's caching allocator, so the
step can be stream-captured once (``torch.cuda.CUDAGraph`` = hipGraph on ROCm) and replayed: same kernels, same
arguments, one submission.  Requirements met by the package: static gradient storage (``parallel.FlatGrads``),
device-resident Adam step count (``optim.HipAdam.step_dev``), device-side noise (``DeviceNoise``: the library's counter-based
generator keeps its draw number in device memory, so every replay draws fresh numbers with no torch generator in the graph),
no host synchronisation inside the step.
"""
import torch


def _detached(out):
    """Outputs without their autograd graph: a graph kept alive from an earlier iteration pins its AccumulateGrad nodes to
    the stream of that iteration, and the next backward would then accumulate outside the stream being captured."""
    if torch.is_tensor(out):
        return out.detach()
    if isinstance(out, (tuple, list)):
        return type(out)(_detached(o) for o in out)
    return out


_capture = {'active': False, 'after': []}


def capturing_step():
    """True while a GraphedStep is capturing: code inside the step may then defer one-time work to after_capture()."""
    return _capture['active']


def after_capture(fn):
    """Run ``fn()`` once, eagerly, right after the running GraphedStep capture ends (before its first replay): for constants
    of the captured graph that are only known during the capture (e.g. a table of the gradient tensors' addresses)."""
    _capture['after'].append(fn)


class GraphedStep:
    def __init__(self, step_fn, generators=(), warmup=3, grad_params=None):
        """step_fn() -> tensor or tuple of tensors (static outputs, overwritten by every replay).
        grad_params: parameters whose ``.grad`` the step (re)binds and a consumer OUTSIDE the graph reads (the data-parallel
        gradient gather / all-reduce between replays).  Capturing rebinds ``p.grad`` to tensors of this graph's pool that only a
        replay fills, and every graph has its own: ``warm_grads`` are the gradients the last eager warm-up step really computed,
        ``grads`` the capture-time tensors; ``__call__`` rebinds ``p.grad`` to them after every replay, ``bind_warm_grads()``
        to the former for the one step the warm-up itself stands for."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        self.warm_out = None
        with torch.cuda.stream(side):
            for _ in range(warmup):      # lazy initialisation (generators, LDS attributes, allocator pools)
                self.warm_out = _detached(step_fn())   # these are real steps; the last one's outputs stay readable here
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.grad_params = list(grad_params) if grad_params is not None else None
        self.warm_grads = [p.grad for p in self.grad_params] if self.grad_params is not None else None
        self.graph = torch.cuda.CUDAGraph()
        for g in generators:
            if g is not None:
                self.graph.register_generator_state(g)
        # thread_local: other threads of the process (e.g. the RCCL watchdog polling its events) may keep calling the
        # runtime while this thread captures
        _capture['active'], _capture['after'] = True, []
        try:
            with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
                self.out = _detached(step_fn())
        finally:
            _capture['active'] = False
        for fn in _capture['after']:
            fn()
        _capture['after'] = []
        torch.cuda.synchronize()
        self.grads = [p.grad for p in self.grad_params] if self.grad_params is not None else None
        self.eager_steps = warmup

    def _bind(self, grads):
        for p, g in zip(self.grad_params, grads):
            p.grad = g

    def bind_warm_grads(self):
        """``p.grad`` := what the last warm-up step computed (valid once: the tensors are released afterwards)."""
        self._bind(self.warm_grads)
        self.warm_grads = None

    def __call__(self):
        self.graph.replay()
        if self.grads is not None:
            self._bind(self.grads)
        return self.out


def device_generators(model):
    """torch.Generator objects on the device reachable from ``model`` (a graph that replays torch-native draws must have them
    registered).  DeviceNoise needs none."""
    from .model.core.noise import DeviceNoise
    gens, seen = [], set()

    def visit(obj):
        if id(obj) in seen:
            return
        seen.add(id(obj))
        if isinstance(obj, DeviceNoise):             # the library's own generator: its state is device memory the kernel advances
            return
        if isinstance(obj, torch.Generator):
            if obj.device.type == 'cuda':
                gens.append(obj)
            return
        if isinstance(obj, torch.nn.Module):
            for v in vars(obj).values():
                visit(v)
            for m in obj.children():
                visit(m)
    visit(model)
    return gens
