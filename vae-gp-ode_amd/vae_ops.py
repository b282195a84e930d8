"""autograd bindings of the conv-VAE HIP kernels (csrc/vae_conv.hip).  Every forward AND backward is a
call into libgpode_hip.so; torch only allocates the tensors and records the graph."""
import ctypes
import os

import torch

from . import _lib, ops
from .ops import _chk, _ptr, _stream


def _new(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def _scratch(n, like):
    return torch.empty(max(int(n), 4), dtype=torch.float32, device=like.device)


def _wgrad_scratch(B, Ci, Co, K, like):
    return _scratch(_lib.load().gpode_conv_wgrad_scratch(B, Ci, Co, K), like)


def _bn_scratch(B, C, like):
    return _scratch(_lib.load().gpode_bn_scratch(B, C), like)


# ---- deferred final reductions (include/gpode.h: gpode_defer_reductions / gpode_flush_reductions) --------------------------------
# set_deferred_reductions(True): the weight / bias gradients and BatchNorm channel sums produced by the backward functions below
# become valid at the END of the backward pass -- ONE launch reduces all their workgroup partials (from autograd's end-of-backward
# callback) instead of one few-microsecond launch per function (14 per step of a first-order model: graph nodes on the critical
# chain).  Sound only when nothing reads those tensors earlier, and autograd does read in two cases: it CLONES an incoming
# gradient that has a second owner, and it ADDS to ``p.grad`` when that exists.  Hence: on only with an optimiser that drops the
# gradients in zero_grad and takes over the tensors autograd hands it (optim.HipAdam(bucketed=False | 'gather'):
# ``allows_deferred_reductions``): creating a HipAdam sets the switch for its kind (the optimiser in charge of the step decides);
# the default, and the setting for any other consumer of ``.grad``, is off.  configs[0]: 14 reduction launches -> 1.
_deferred = {'on': False, 'keep': [], 'queued': False}
_DEFER_ALLOWED = os.environ.get('GPODE_EAGER_REDUCTIONS', '0') != '1'


def set_deferred_reductions(on):
    _deferred['on'] = bool(on) and _DEFER_ALLOWED


def _flush_deferred():
    _deferred['queued'] = False
    try:
        _lib.call('gpode_flush_reductions', _stream())
    finally:
        _deferred['keep'].clear()


def _bwd_call(name, *args, keep=()):
    """A backward-pass entry point whose last step is a reduction of workgroup partials (``keep``: its scratch buffers)."""
    if not (_deferred['on'] and torch._C._current_graph_task_id() >= 0):
        return _lib.call(name, *args)
    lib = _lib.load()
    if not _deferred['queued']:
        lib.gpode_defer_reductions(2)                # drop leftovers of an aborted pass, then record
        torch.autograd.Variable._execution_engine.queue_callback(_flush_deferred)
        _deferred['queued'] = True
    else:
        lib.gpode_defer_reductions(1)
    try:
        _lib.call(name, *args)
    finally:
        lib.gpode_defer_reductions(0)
    _deferred['keep'].extend(keep)


_bn_sync = None        # parallel.BatchNormSync while data-parallel training normalises with global-minibatch statistics


def set_bn_sync(group):
    """group: parallel.BatchNormSync (or None = per-process statistics, the single-GPU path)."""
    global _bn_sync
    _bn_sync = group


def _bn_global_stats(x, gamma, beta, running_mean, running_var, nbt, momentum, eps, scratch):
    """Forward half of the cross-rank BatchNorm: local moments -> all-gather -> rank-ordered combination.
    Returns (mean, invstd, table[C][4])."""
    B, C = x.shape[0], x.shape[1]
    HW = x[0, 0].numel()
    mom = _new((2 * C + 1,), x)
    _lib.call('gpode_bn_moments', _ptr(x), _ptr(mom), B, C, HW, _ptr(scratch), _stream())
    gathered = _bn_sync.gather(mom)
    mean, invstd, table = _new((C,), x), _new((C,), x), _new((C, 4), x)
    _lib.call('gpode_bn_finalize', _ptr(gathered), _bn_sync.world, _ptr(_chk(gamma, 'gamma')), _ptr(_chk(beta, 'beta')), _ptr(mean),
              _ptr(invstd), _ptr(running_mean), _ptr(running_var), _ptr(nbt), ctypes.c_float(momentum), ctypes.c_float(eps), _ptr(table), C,
              _stream())
    return mean, invstd, table


def _bn_global_bwd(sync, x, gy, gamma, beta, mean, invstd, relu):
    """Backward half: local sums -> all-gather -> gx with the global centring terms.  Returns (gx, ggamma, gbeta, chansum)."""
    B, C = x.shape[0], x.shape[1]
    HW = x[0, 0].numel()
    scratch = _bn_scratch(B, C, x)
    sums = _new((2 * C,), x)
    gy = gy.contiguous()
    _lib.call('gpode_bn_bwd_sums', _ptr(x), _ptr(gy), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(invstd), _ptr(sums), B, C, HW, int(relu),
              _ptr(scratch), _stream())
    gathered = sync.gather(sums)
    gx, gg, gb, cs = _new(x.shape, x), _new((C,), x), _new((C,), x), _new((C,), x)
    _bwd_call('gpode_bn_bwd_apply', _ptr(x), _ptr(gy), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(invstd), _ptr(gathered),
              _ptr(sync.weights(x.device)), sync.world, ctypes.c_float(sync.count_all(B * HW)), _ptr(gx), _ptr(gg), _ptr(gb), _ptr(cs),
              B, C, HW, int(relu), _ptr(scratch), _stream(), keep=(scratch,))
    return gx, gg, gb, cs


# -- two independent BatchNorm layers of the same depth (the position and the velocity encoder of a second-order model, vae.py:14-19)
# with ONE all-gather per direction instead of two: on N ranks every collective of a step is a latency-bound exchange of a few
# hundred bytes, so their number is what counts (parallel.BatchNormSync.gather_many; switch: set_pack_bn_gathers) ---------------
_pack_bn = {'on': os.environ.get('GPODE_PACK_BN_GATHERS', '0') == '1'}


def set_pack_bn_gathers(on):
    _pack_bn['on'] = bool(on)


def pack_bn_gathers():
    return _pack_bn['on'] and _bn_sync is not None


class _BatchNormTrainPair(torch.autograd.Function):
    """(BatchNorm2d_train + ReLU)(x1), (BatchNorm2d_train + ReLU)(x2) under cross-rank statistics, the two layers' moments (forward)
    and backward sums (backward) exchanged in one packed all-gather each.  Same kernels, same rank order of every combination as two
    _BatchNormTrain calls: results are bit-identical to the unpacked form."""

    @staticmethod
    def forward(ctx, x1, g1, b1, rm1, rv1, nbt1, x2, g2, b2, rm2, rv2, nbt2, momentum, eps, relu):
        sync = _bn_sync
        xs, moms = (_chk(x1, 'x1'), _chk(x2, 'x2')), []
        for x in xs:
            B, C, HW = x.shape[0], x.shape[1], x[0, 0].numel()
            mom = _new((2 * C + 1,), x)
            _lib.call('gpode_bn_moments', _ptr(x), _ptr(mom), B, C, HW, _ptr(_bn_scratch(B, C, x)), _stream())
            moms.append(mom)
        gathered = sync.gather_many(moms)
        ys, saved = [], []
        for x, gam, bet, rm, rv, nbt, got in zip(xs, (g1, g2), (b1, b2), (rm1, rm2), (rv1, rv2), (nbt1, nbt2), gathered):
            B, C, HW = x.shape[0], x.shape[1], x[0, 0].numel()
            mean, invstd, table, y = _new((C,), x), _new((C,), x), _new((C, 4), x), _new(x.shape, x)
            _lib.call('gpode_bn_finalize', _ptr(got), sync.world, _ptr(_chk(gam, 'gamma')), _ptr(_chk(bet, 'beta')), _ptr(mean), _ptr(invstd),
                      _ptr(rm), _ptr(rv), _ptr(nbt), ctypes.c_float(momentum), ctypes.c_float(eps), _ptr(table), C, _stream())
            _lib.call('gpode_bn_apply', _ptr(x), _ptr(table), _ptr(y), B, C, HW, int(relu), _stream())
            ys.append(y)
            saved += [x, gam, bet, mean, invstd]
        ctx.save_for_backward(*saved)
        ctx.sync, ctx.relu = sync, int(relu)
        return ys[0], ys[1]

    @staticmethod
    def backward(ctx, gy1, gy2):
        sv, sync = ctx.saved_tensors, ctx.sync
        packs, sums, scr = [], [], []
        for i, gy in enumerate((gy1, gy2)):
            x, gam, bet, mean, invstd = sv[5 * i:5 * i + 5]
            B, C, HW = x.shape[0], x.shape[1], x[0, 0].numel()
            gy = gy.contiguous()
            scratch, s = _bn_scratch(B, C, x), _new((2 * C,), x)
            _lib.call('gpode_bn_bwd_sums', _ptr(x), _ptr(gy), _ptr(gam), _ptr(bet), _ptr(mean), _ptr(invstd), _ptr(s), B, C, HW, ctx.relu,
                      _ptr(scratch), _stream())
            packs.append((x, gy, gam, bet, mean, invstd))
            sums.append(s)
            scr.append(scratch)
        gathered = sync.gather_many(sums)
        out = []
        for (x, gy, gam, bet, mean, invstd), got, scratch in zip(packs, gathered, scr):
            B, C, HW = x.shape[0], x.shape[1], x[0, 0].numel()
            gx, gg, gb, cs = _new(x.shape, x), _new((C,), x), _new((C,), x), _new((C,), x)
            _bwd_call('gpode_bn_bwd_apply', _ptr(x), _ptr(gy), _ptr(gam), _ptr(bet), _ptr(mean), _ptr(invstd), _ptr(got),
                      _ptr(sync.weights(x.device)), sync.world, ctypes.c_float(sync.count_all(B * HW)), _ptr(gx), _ptr(gg), _ptr(gb), _ptr(cs),
                      B, C, HW, ctx.relu, _ptr(scratch), _stream(), keep=(scratch,))
            gx._gpode_chansum = cs
            out.append((gx, gg, gb))
        (gx1, gg1, gb1), (gx2, gg2, gb2) = out
        return gx1, gg1, gb1, None, None, None, gx2, gg2, gb2, None, None, None, None, None, None


def batch_norm_train_pair(x1, bn1, x2, bn2, relu):
    """Two training-mode BatchNorm2d modules on their own inputs with packed cross-rank exchanges (see _BatchNormTrainPair)."""
    return _BatchNormTrainPair.apply(x1, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, bn1.num_batches_tracked,
                                     x2, bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var, bn2.num_batches_tracked,
                                     bn1.momentum, bn1.eps, relu)


_dec10_fused = os.environ.get('GPODE_DEC10_BN_UNFUSED', '0') != '1'


_DEC10_WGRAD_FUSED = os.environ.get('GPODE_DEC10_WGRAD_PASS', '0') != '1'


def _dec10_bn_bwd(sync, c, gy, w, gamma, beta, mean, invstd, gw=None, gbias=None):
    """decnn.10's input gradient + the BatchNorm/ReLU backward in front of it in two passes over c (include/gpode.h,
    gpode_dec10_bn_bwd_*): (gc, ggamma, gbeta, channel sums of gc).  ``gw`` (a tensor to fill): the layer's weight gradient rides
    in the first pass (gpode_dec10_bn_bwd_sums_wgrad)."""
    B = c.shape[0]
    scratch = _scratch(_lib.load().gpode_dec10_bn_scratch_floats(), c)
    gc, gg, gb, cs = _new(c.shape, c), _new((16,), c), _new((16,), c), _new((16,), c)
    head = (_ptr(c), _ptr(gy), _ptr(w), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(invstd))
    sums = _new((32,), c) if sync is not None else None
    if gw is not None:
        ws = _scratch(_lib.load().gpode_dec10_bn_wgrad_scratch_floats(), c)
        _bwd_call('gpode_dec10_bn_bwd_sums_wgrad', *head, _ptr(sums), _ptr(gw), _ptr(gbias), B, _ptr(scratch), _ptr(ws), _stream(),
                  keep=(ws, scratch))
    else:
        _lib.call('gpode_dec10_bn_bwd_sums', *head, _ptr(sums), B, _ptr(scratch), _stream())
    if sync is None:
        _bwd_call('gpode_dec10_bn_bwd_apply', *head, _ptr(None), _ptr(None), 0, ctypes.c_float(0.0), _ptr(gc), _ptr(gg), _ptr(gb), _ptr(cs),
                  B, _ptr(scratch), _stream(), keep=(scratch,))
    else:
        gathered = sync.gather(sums)
        _bwd_call('gpode_dec10_bn_bwd_apply', *head, _ptr(gathered), _ptr(sync.weights(c.device)), sync.world,
                  ctypes.c_float(sync.count_all(B * 784)), _ptr(gc), _ptr(gg), _ptr(gb), _ptr(cs), B, _ptr(scratch), _stream(), keep=(scratch,))
    return gc, gg, gb, cs


fused_bias_grads = 0   # how many bias gradients arrived ready-made from a BatchNorm backward (see _BatchNormTrain.backward)


def _fused_chansum(gy, C):
    global fused_bias_grads
    cs = getattr(gy, '_gpode_chansum', None)
    if cs is None or tuple(cs.shape) != (C,):
        return None
    # hand the tensor over with no second owner: autograd takes a gradient as it is only when nothing else refers to it, and clones
    # it otherwise -- a read, which must not happen before the deferred reductions have run (set_deferred_reductions)
    del gy._gpode_chansum
    fused_bias_grads += 1
    return cs


def _batch_strided(x):
    """x (B,C,H,W) with dense images at a constant distance (e.g. X[:, 0] of a minibatch X (N,T,1,28,28), odegpvae.py:55-63) and a
    channel count the generic convolution kernels take: floats between images, else 0."""
    if x.dim() != 4 or x.is_contiguous() or x.dtype != torch.float32 or not x.is_cuda or x.shape[1] % 4 == 0 or x.shape[0] < 2:
        return 0
    B, C, H, W = x.shape
    if x.stride()[1:] != (H * W, W, 1) or x.stride(0) < C * H * W:
        return 0
    return x.stride(0)


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, pad):
        xbs = _batch_strided(x)                      # read in place: no contiguous copy of the slice (a launch on the step's first chain)
        if not xbs:
            x = _chk(x, 'x')
        w = _chk(w, 'weight')
        B, Ci, H, W = x.shape
        Co, _, K, _ = w.shape
        Ho, Wo = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
        y = _new((B, Co, Ho, Wo), x)
        if xbs:
            _lib.call('gpode_conv2d_fwd_bs', _ptr(x), xbs, _ptr(w), _ptr(b), _ptr(y), B, Ci, H, W, Co, K, stride, pad, Ho, Wo, _stream())
        else:
            _lib.call('gpode_conv2d_fwd', _ptr(x), _ptr(w), _ptr(b), _ptr(y), B, Ci, H, W, Co, K, stride, pad, Ho, Wo, _stream())
        ctx.save_for_backward(x, w)
        ctx.geom = (B, Ci, H, W, Co, K, stride, pad, Ho, Wo, b is not None)
        ctx.xbs = xbs
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        B, Ci, H, W, Co, K, S, P, Ho, Wo, has_b = ctx.geom
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _new(x.shape, x)
            _lib.call('gpode_conv2d_bwd_data', _ptr(gy), _ptr(w), _ptr(None), _ptr(gx), B, Ci, H, W, Co, K, S, P, Ho, Wo, _stream())
        if ctx.needs_input_grad[1]:
            gw = _new(w.shape, x)
            pre = _fused_chansum(gy, Co) if has_b else None     # bias gradient already produced by the BatchNorm backward
            gb = _new((Co,), x) if (has_b and pre is None) else None
            ws = _wgrad_scratch(B, Ci, Co, K, x)
            if ctx.xbs:
                _bwd_call('gpode_conv2d_bwd_weight_bs', _ptr(x), ctx.xbs, _ptr(gy), _ptr(gw), _ptr(gb), _ptr(ws),
                          B, Ci, H, W, Co, K, S, P, Ho, Wo, _stream(), keep=(ws, gy))
            else:
                _bwd_call('gpode_conv2d_bwd_weight', _ptr(x), _ptr(gy), _ptr(gw), _ptr(gb), _ptr(ws),
                          B, Ci, H, W, Co, K, S, P, Ho, Wo, _stream(), keep=(ws, gy))
            if pre is not None:
                gb = pre
        return gx, gw, gb, None, None


# ---- BatchNorm statistics produced by the convolution in front of it (include/gpode.h: gpode_convT_fwd_stats) ----------------------
# The decoder's transposed convolutions decnn.1/4/7 are each followed by a BatchNorm2d in training mode (vae.py:107-120).  When the
# caller names that module (``stats_for``), the convolution sums the statistics of its output while storing it and its last workgroup
# finalises them: the output tensor then carries (mean, invstd, table) as ``_gpode_bnstats`` and the consuming _BnReluConvT takes them
# instead of launching the statistics pass + table kernel (two graph nodes and one read of the tensor per layer).
_STATS_GEOMS = {(32, 64, 3, 1, 0, 4), (64, 32, 5, 2, 1, 6), (32, 16, 5, 2, 1, 13)}      # (Cin, Cout, K, stride, pad, Hi) of decnn.1/4/7
_fused_stats = os.environ.get('GPODE_BN_STATS_PASS', '0') != '1'
_FUSED_STATS_MIN_IMAGES = int(os.environ.get('GPODE_BN_STATS_FUSED_MIN', '512'))
_slots = {'next': 0}


def _stats_slot(bn):
    s = getattr(bn, '_gpode_slot', None)
    if s is None:
        s = bn._gpode_slot = _slots['next'] % 64
        _slots['next'] += 1
    return s


def _can_fuse_stats(bn, x, Cin, Cout, K, stride, pad, Hi):
    # from 512 images on: below that the statistics pass is a few microseconds and the per-workgroup hand-over at the end of the
    # convolution costs as much.  configs[0] (512 images): 0.722 vs 0.724 ms per step fused vs separate -- even, with five graph nodes
    # less; configs[1] (4096): 2.62 vs 2.66.  (With four loads in flight in the last workgroup's combine, round 3's first version, the
    # break-even was at 1024 images: bn_sink.hpp.)
    return (_fused_stats and bn is not None and bn.training and _bn_sync is None and (Cin, Cout, K, stride, pad, Hi) in _STATS_GEOMS
            and x.shape[0] >= _FUSED_STATS_MIN_IMAGES and x.data_ptr() % 16 == 0 and os.environ.get('GPODE_CONV_VALU', '0') != '1')


def _convT_fwd_stats(x, table_in, w, b, y, geom, bn):
    """y = convT(x [after BatchNorm + ReLU by table_in], w) + b and the statistics of y for the BatchNorm module ``bn`` behind it."""
    B, Cout, Ht, Wt, Cin, K, stride, pad, Hi, Wi = geom
    mean, invstd, table = _new((Cout,), x), _new((Cout,), x), _new((Cout, 4), x)
    scratch = _scratch(_lib.load().gpode_convT_fwd_stats_scratch(Cout), x)
    _lib.call('gpode_convT_fwd_stats', _ptr(x), _ptr(table_in), _ptr(w), _ptr(b), _ptr(y), B, Cout, Ht, Wt, Cin, K, stride, pad, Hi, Wi,
              _ptr(_chk(bn.weight, 'gamma')), _ptr(_chk(bn.bias, 'beta')), _ptr(mean), _ptr(invstd), _ptr(bn.running_mean),
              _ptr(bn.running_var), _ptr(bn.num_batches_tracked), ctypes.c_float(bn.momentum), ctypes.c_float(bn.eps), _ptr(table),
              _ptr(scratch), _stats_slot(bn), _stream())
    y._gpode_bnstats = (bn, mean, invstd, table)


class _ConvT2d(torch.autograd.Function):
    """nn.ConvTranspose2d as the adjoint of the convolution that shares its weight buffer."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, out_pad, stats_for=None):
        x, w = _chk(x, 'x'), _chk(w, 'weight')
        B, Cin, Hi, Wi = x.shape
        _, Cout, K, _ = w.shape
        Ht, Wt = (Hi - 1) * stride - 2 * pad + K + out_pad, (Wi - 1) * stride - 2 * pad + K + out_pad
        y = _new((B, Cout, Ht, Wt), x)
        # conv geometry: "input" (B,Ci=Cout,H=Ht,W=Wt), "output" (B,Co=Cin,Ho=Hi,Wo=Wi)
        if _can_fuse_stats(stats_for, x, Cin, Cout, K, stride, pad, Hi):
            _convT_fwd_stats(x, None, w, b, y, (B, Cout, Ht, Wt, Cin, K, stride, pad, Hi, Wi), stats_for)
        else:
            _lib.call('gpode_conv2d_bwd_data', _ptr(x), _ptr(w), _ptr(b), _ptr(y), B, Cout, Ht, Wt, Cin, K, stride, pad, Hi, Wi, _stream())
        ctx.save_for_backward(x, w)
        ctx.geom = (B, Cout, Ht, Wt, Cin, K, stride, pad, Hi, Wi, b is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        B, Cout, Ht, Wt, Cin, K, S, P, Hi, Wi, has_b = ctx.geom
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _new(x.shape, x)
            _lib.call('gpode_conv2d_fwd', _ptr(gy), _ptr(w), _ptr(None), _ptr(gx), B, Cout, Ht, Wt, Cin, K, S, P, Hi, Wi, _stream())
        if ctx.needs_input_grad[1]:
            gw = _new(w.shape, x)
            ws = _wgrad_scratch(B, Cout, Cin, K, x)
            _bwd_call('gpode_conv2d_bwd_weight', _ptr(gy), _ptr(x), _ptr(gw), _ptr(None), _ptr(ws),
                      B, Cout, Ht, Wt, Cin, K, S, P, Hi, Wi, _stream(), keep=(ws,))
            if has_b:
                gb = _fused_chansum(gy, Cout)
                if gb is None:
                    gb, bs = _new((Cout,), x), _bn_scratch(B, Cout, x)
                    _bwd_call('gpode_chan_sum', _ptr(gy), _ptr(gb), B, Cout, Ht * Wt, _ptr(bs), _stream(), keep=(bs,))
        return gx, gw, gb, None, None, None, None


class _BnReluConvT(torch.autograd.Function):
    """ConvTranspose2d(ReLU(BatchNorm2d_train(c))) (vae.py:113-120, one decoder stage) with the normalised activation never
    written to memory: forward = batch statistics of c (one read) + the transposed convolution applying normalise / affine /
    ReLU while it stages its input; backward = weight gradient (same staging), input gradient of the convolution, then the
    BatchNorm backward on c.  Training mode only; the eval-mode path keeps the separate ops."""

    @staticmethod
    def forward(ctx, c, gamma, beta, running_mean, running_var, nbt, momentum, eps, w, b, stride, pad, out_pad, bn=None, stats_for=None):
        pre = getattr(c, '_gpode_bnstats', None)     # the convolution that produced c summed its statistics for module ``bn`` already
        c, w = _chk(c, 'c'), _chk(w, 'weight')
        B, Cin, Hi, Wi = c.shape
        _, Cout, K, _ = w.shape
        Ht, Wt = (Hi - 1) * stride - 2 * pad + K + out_pad, (Wi - 1) * stride - 2 * pad + K + out_pad
        ctx.sync = _bn_sync
        if pre is not None and pre[0] is bn and bn is not None and _bn_sync is None:
            _, mean, invstd, table = pre
        elif _bn_sync is not None:
            mean, invstd, table = _bn_global_stats(c, gamma, beta, running_mean, running_var, nbt, momentum, eps, _bn_scratch(B, Cin, c))
        else:
            mean, invstd, table = _new((Cin,), c), _new((Cin,), c), _new((Cin, 4), c)
            _lib.call('gpode_bn_stats', _ptr(c), _ptr(_chk(gamma, 'gamma')), _ptr(_chk(beta, 'beta')), _ptr(mean), _ptr(invstd),
                      _ptr(running_mean), _ptr(running_var), _ptr(nbt), ctypes.c_float(momentum), ctypes.c_float(eps), _ptr(table),
                      B, Cin, Hi * Wi, _ptr(_bn_scratch(B, Cin, c)), _stream())
        y = _new((B, Cout, Ht, Wt), c)
        if _can_fuse_stats(stats_for, c, Cin, Cout, K, stride, pad, Hi):
            _convT_fwd_stats(c, table, w, b, y, (B, Cout, Ht, Wt, Cin, K, stride, pad, Hi, Wi), stats_for)
        else:
            _lib.call('gpode_conv2d_bwd_data_bn', _ptr(c), _ptr(table), _ptr(w), _ptr(b), _ptr(y), B, Cout, Ht, Wt, Cin, K, stride, pad, Hi, Wi,
                      _stream())
        ctx.save_for_backward(c, gamma, beta, mean, invstd, table, w)
        ctx.geom = (B, Cout, Ht, Wt, Cin, K, stride, pad, Hi, Wi, b is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        c, gamma, beta, mean, invstd, table, w = ctx.saved_tensors
        B, Cout, Ht, Wt, Cin, K, S, P, Hi, Wi, has_b = ctx.geom
        ops.side_heartbeat()
        gy = gy.contiguous()
        gw = gb = None
        last_stage = _dec10_fused and (Cout, Cin, K, S, P, Hi, Wi, Ht, Wt) == (1, 16, 5, 1, 2, 28, 28, 28, 28)
        gw_rides = last_stage and _DEC10_WGRAD_FUSED and ctx.needs_input_grad[8] and c.data_ptr() % 16 == 0
        if ctx.needs_input_grad[8]:
            gw = _new(w.shape, c)
            if not gw_rides:                         # (the last stage's weight gradient rides in its BatchNorm sums pass below)
                ws = _wgrad_scratch(B, Cout, Cin, K, c)
                _bwd_call('gpode_conv2d_bwd_weight_bn', _ptr(gy), _ptr(c), _ptr(table), _ptr(gw), _ptr(None),
                          _ptr(ws), B, Cout, Ht, Wt, Cin, K, S, P, Hi, Wi, _stream(), keep=(ws,))
            if has_b:
                gb = _fused_chansum(gy, Cout)
                if gb is None:
                    gb = _new((Cout,), c)
                    if not gw_rides:                 # (with the weight gradient, the bias gradient rides in the sums pass too)
                        bs = _bn_scratch(B, Cout, c)
                        _bwd_call('gpode_chan_sum', _ptr(gy), _ptr(gb), B, Cout, Ht * Wt, _ptr(bs), _stream(), keep=(bs,))
                        gb_rides = None
                    else:
                        gb_rides = gb
                else:
                    gb_rides = None
            else:
                gb_rides = None
        else:
            gb_rides = None
        if last_stage:
            # the decoder's last stage: the gradient w.r.t. the normalised activation is recomputed inside both BatchNorm passes
            gc, gg, gbeta, cs = _dec10_bn_bwd(ctx.sync, c, gy, w, gamma, beta, mean, invstd, gw=gw if gw_rides else None, gbias=gb_rides)
            gc._gpode_chansum = cs
            return gc, gg, gbeta, None, None, None, None, None, gw, gb, None, None, None, None, None
        # gradient w.r.t. the (never materialised) normalised activation, then through the BatchNorm to c
        ga = _new(c.shape, c)
        _lib.call('gpode_conv2d_fwd', _ptr(gy), _ptr(w), _ptr(None), _ptr(ga), B, Cout, Ht, Wt, Cin, K, S, P, Hi, Wi, _stream())
        if ctx.sync is not None:
            gc, gg, gbeta, cs = _bn_global_bwd(ctx.sync, c, ga, gamma, beta, mean, invstd, 1)
        else:
            gc, gg, gbeta, cs = _new(c.shape, c), _new((Cin,), c), _new((Cin,), c), _new((Cin,), c)
            bs = _bn_scratch(B, Cin, c)
            _bwd_call('gpode_bn_bwd', _ptr(c), _ptr(ga), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(invstd), _ptr(gc), _ptr(gg), _ptr(gbeta),
                      _ptr(cs), B, Cin, Hi * Wi, 1, _ptr(bs), _stream(), keep=(bs,))
        gc._gpode_chansum = cs
        return gc, gg, gbeta, None, None, None, None, None, gw, gb, None, None, None, None, None


class _BatchNormTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, relu, nbt=None):
        x = _chk(x, 'x')
        B, C = x.shape[0], x.shape[1]
        HW = x[0, 0].numel()
        y = _new(x.shape, x)
        mean, invstd = _new((C,), x), _new((C,), x)
        if nbt is not None and not (nbt.is_cuda and nbt.dtype == torch.int64):
            raise _lib.GpodeError('num_batches_tracked must be an int64 CUDA/HIP scalar')
        ctx.sync = _bn_sync
        if _bn_sync is not None:
            mean, invstd, table = _bn_global_stats(x, gamma, beta, running_mean, running_var, nbt, momentum, eps, _bn_scratch(B, C, x))
            _lib.call('gpode_bn_apply', _ptr(x), _ptr(table), _ptr(y), B, C, HW, int(relu), _stream())
        else:
            _lib.call('gpode_bn_fwd', _ptr(x), _ptr(_chk(gamma, 'gamma')), _ptr(_chk(beta, 'beta')), _ptr(y), _ptr(mean), _ptr(invstd),
                      _ptr(running_mean), _ptr(running_var), _ptr(nbt), ctypes.c_float(momentum), ctypes.c_float(eps), B, C, HW, int(relu),
                      _ptr(_bn_scratch(B, C, x)), _stream())
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        ctx.relu = int(relu)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        B, C = x.shape[0], x.shape[1]
        HW = x[0, 0].numel()
        if ctx.sync is not None:
            gx, gg, gb, cs = _bn_global_bwd(ctx.sync, x, gy, gamma, beta, mean, invstd, ctx.relu)
        else:
            gx, gg, gb, cs = _new(x.shape, x), _new((C,), x), _new((C,), x), _new((C,), x)
            bs = _bn_scratch(B, C, x)
            _bwd_call('gpode_bn_bwd', _ptr(x), _ptr(gy.contiguous()), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(invstd), _ptr(gx), _ptr(gg),
                      _ptr(gb), _ptr(cs), B, C, HW, ctx.relu, _ptr(bs), _stream(), keep=(bs,))
        # the channel sums of gx ride along with it: the convolution that produced x needs exactly these as its bias gradient
        gx._gpode_chansum = cs
        return gx, gg, gb, None, None, None, None, None, None


class _BatchNormEval(torch.autograd.Function):
    """nn.BatchNorm2d under module.eval(): running statistics, gradient w.r.t. the input only."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, relu):
        x = _chk(x, 'x')
        B, C = x.shape[0], x.shape[1]
        HW = x[0, 0].numel()
        y = _new(x.shape, x)
        _lib.call('gpode_bn_eval', _ptr(x), _ptr(None), _ptr(_chk(gamma, 'gamma')), _ptr(_chk(beta, 'beta')), _ptr(running_mean),
                  _ptr(running_var), ctypes.c_float(eps), _ptr(y), B, C, HW, int(relu), _stream())
        ctx.save_for_backward(x, gamma, beta, running_mean, running_var)
        ctx.eps, ctx.relu = eps, int(relu)
        return y

    @staticmethod
    def backward(ctx, gy):
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise _lib.GpodeError('eval-mode BatchNorm is built for frozen layers (main.py:160-161 sets requires_grad=False)')
        x, gamma, beta, rm, rv = ctx.saved_tensors
        B, C = x.shape[0], x.shape[1]
        HW = x[0, 0].numel()
        gx = _new(x.shape, x)
        _lib.call('gpode_bn_eval', _ptr(x), _ptr(gy.contiguous()), _ptr(gamma), _ptr(beta), _ptr(rm), _ptr(rv), ctypes.c_float(ctx.eps),
                  _ptr(gx), B, C, HW, ctx.relu, _stream())
        return gx, None, None, None, None, None, None


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode):
        x = _chk(x, 'x')
        y = _new(x.shape, x)
        _lib.call('gpode_act_fwd', _ptr(x), _ptr(y), x.numel(), mode, _stream())
        ctx.save_for_backward(y)
        ctx.mode = mode
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gx = _new(y.shape, y)
        _lib.call('gpode_act_bwd', _ptr(y), _ptr(gy.contiguous()), _ptr(gx), y.numel(), ctx.mode, _stream())
        return gx, None


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        x, w = _chk(x, 'x'), _chk(w, 'weight')
        B, In = x.shape
        Out = w.shape[0]
        y = _new((B, Out), x)
        _lib.call('gpode_linear_fwd', _ptr(x), _ptr(w), _ptr(b), _ptr(y), B, In, Out, _stream())
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        B, In = x.shape
        Out = w.shape[0]
        gx = _new(x.shape, x) if ctx.needs_input_grad[0] else None
        gw = _new(w.shape, x) if ctx.needs_input_grad[1] else None
        gb = _new((Out,), x) if (ctx.has_b and gw is not None) else None
        ws = _scratch(_lib.load().gpode_linear_bwd_scratch(B, In, Out), x) if (gw is not None and B >= 1024 and In <= 8) else None
        _bwd_call('gpode_linear_bwd', _ptr(x), _ptr(w), _ptr(gy.contiguous()), _ptr(gx), _ptr(gw), _ptr(gb), B, In, Out, _ptr(ws), _stream(),
                  keep=(ws,) if ws is not None else ())
        return gx, gw, gb


class _LinearReluIn(torch.autograd.Function):
    """linear(relu(x), w, b) with the ReLU folded into the layer's loads in both directions (gpode_linear_relu_fwd / _bwd): the
    encoder's Conv2d -> ReLU -> Flatten -> Linear (vae.py:58-61, 72-74) without the activation tensor and its two launches."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w = _chk(x, 'x'), _chk(w, 'weight')
        B, In = x.shape
        Out = w.shape[0]
        y = _new((B, Out), x)
        _lib.call('gpode_linear_relu_fwd', _ptr(x), _ptr(w), _ptr(b), _ptr(y), B, In, Out, _stream())
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        B, In = x.shape
        Out = w.shape[0]
        gx = _new(x.shape, x) if ctx.needs_input_grad[0] else None
        gw = _new(w.shape, x) if ctx.needs_input_grad[1] else None
        gb = _new((Out,), x) if (ctx.has_b and gw is not None) else None
        _lib.call('gpode_linear_relu_bwd', _ptr(x), _ptr(w), _ptr(gy.contiguous()), _ptr(gx), _ptr(gw), _ptr(gb), B, In, Out, _stream())
        return gx, gw, gb


class _LogLik(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, z):
        X, z = _chk(X, 'X'), _chk(z, 'z')
        ll = _new(z.shape, z)
        _lib.call('gpode_loglik_fwd', _ptr(X), _ptr(z), _ptr(ll), z.numel(), X.numel(), _stream())
        ctx.save_for_backward(X, z)
        return ll

    @staticmethod
    def backward(ctx, g):
        X, z = ctx.saved_tensors
        gz = _new(z.shape, z)
        _lib.call('gpode_loglik_bwd', _ptr(X), _ptr(z), _ptr(g.contiguous()), _ptr(gz), z.numel(), X.numel(), _stream())
        return None, gz


class _LogLikRowSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, z, rows):
        X, z = _chk(X, 'X'), _chk(z, 'z')
        inner = z.numel() // rows
        out = _new((rows,), z)
        _lib.call('gpode_loglik_rowsum_fwd', _ptr(X), _ptr(z), _ptr(out), rows, inner, X.numel(), _stream())
        ctx.save_for_backward(X, z)
        ctx.rows, ctx.inner = rows, inner
        return out

    @staticmethod
    def backward(ctx, g):
        X, z = ctx.saved_tensors
        gz = _new(z.shape, z)
        _lib.call('gpode_loglik_rowsum_bwd', _ptr(X), _ptr(z), _ptr(g.contiguous()), _ptr(gz), ctx.rows, ctx.inner, X.numel(), _stream())
        return None, gz, None


def _packed_halves(mu, logvar):
    """If (mu, logvar) are the two column halves of one contiguous (N, 2q) tensor (Encoder.forward returns fc(h).chunk(2)),
    return that tensor: the kernels then read both halves in place and produce ONE gradient for it."""
    base = getattr(mu, '_base', None)
    if base is None or getattr(logvar, '_base', None) is not base or base.dim() != 2 or not base.is_contiguous():
        return None
    N, q = mu.shape
    if tuple(base.shape) != (N, 2 * q) or mu.stride() != (2 * q, 1) or logvar.stride() != (2 * q, 1):
        return None
    if mu.data_ptr() != base.data_ptr() or logvar.data_ptr() != base.data_ptr() + 4 * q:
        return None
    return base


class _Reparam(torch.autograd.Function):
    """z = mu + exp(logvar / 2) * eps (vae.py:75-78) in one launch; h = [mu | logvar] packed (N, 2q)."""

    @staticmethod
    def forward(ctx, h, eps):
        h = _chk(h, 'h')
        N, q = h.shape[0], h.shape[1] // 2
        eps = _chk(eps, 'eps', (N, q))
        z = _new((N, q), h)
        _lib.call('gpode_reparam_fwd', _ptr(h), ctypes.c_void_p(h.data_ptr() + 4 * q), 2 * q, _ptr(eps), _ptr(z), N, q, _stream())
        ctx.save_for_backward(h, eps)
        return z

    @staticmethod
    def backward(ctx, gz):
        h, eps = ctx.saved_tensors
        N, q = h.shape[0], h.shape[1] // 2
        g = _new(h.shape, h)
        _lib.call('gpode_reparam_bwd', _ptr(gz.contiguous()), ctypes.c_void_p(h.data_ptr() + 4 * q), 2 * q, _ptr(eps), _ptr(g),
                  ctypes.c_void_p(g.data_ptr() + 4 * q), 2 * q, N, q, _stream())
        return g, None


class _ReparamKL(torch.autograd.Function):
    """_Reparam that also returns the partial sums of KL(q(z0) || N(0, I)) over its workgroups (gpode_reparam_kl_fwd): the ELBO adds
    them up instead of reading (mu | logvar) again, and the backward writes the gradient of h = [mu | logvar] through z AND through
    the KL sum in one launch -- otherwise two gradient tensors that autograd adds in a launch of its own."""

    @staticmethod
    def forward(ctx, h, eps):
        h = _chk(h, 'h')
        N, q = h.shape[0], h.shape[1] // 2
        eps = _chk(eps, 'eps', (N, q))
        z, klpart = _new((N, q), h), _new(((N * q + 255) // 256,), h)
        _lib.call('gpode_reparam_kl_fwd', _ptr(h), ctypes.c_void_p(h.data_ptr() + 4 * q), 2 * q, _ptr(eps), _ptr(z), _ptr(klpart), N, q, _stream())
        ctx.save_for_backward(h, eps)
        ctx.set_materialize_grads(False)
        return z, klpart

    @staticmethod
    def backward(ctx, gz, gkl):
        h, eps = ctx.saved_tensors
        N, q = h.shape[0], h.shape[1] // 2
        g = _new(h.shape, h)
        _lib.call('gpode_reparam_kl_bwd', _ptr(gz.contiguous() if gz is not None else None), _ptr(gkl.contiguous() if gkl is not None else None),
                  _ptr(h), ctypes.c_void_p(h.data_ptr() + 4 * q), 2 * q, _ptr(eps), _ptr(g), ctypes.c_void_p(g.data_ptr() + 4 * q), 2 * q, N, q,
                  _stream())
        return g, None


class _NormalKL(torch.autograd.Function):
    """sum_d KL(N(mu, exp(logvar/2)) || N(0, 1)) per row (what kl_divergence(q_dist, prior).sum(-1) evaluates,
    create_model.py:47-49); h = [mu | logvar] packed (N, 2q)."""

    @staticmethod
    def forward(ctx, h):
        h = _chk(h, 'h')
        N, q = h.shape[0], h.shape[1] // 2
        out = _new((N,), h)
        _lib.call('gpode_normal_kl_fwd', _ptr(h), ctypes.c_void_p(h.data_ptr() + 4 * q), 2 * q, _ptr(out), N, q, _stream())
        ctx.save_for_backward(h)
        return out

    @staticmethod
    def backward(ctx, g):
        (h,) = ctx.saved_tensors
        N, q = h.shape[0], h.shape[1] // 2
        gh = _new(h.shape, h)
        _lib.call('gpode_normal_kl_bwd', _ptr(g.contiguous()), _ptr(h), ctypes.c_void_p(h.data_ptr() + 4 * q), 2 * q, _ptr(gh),
                  ctypes.c_void_p(gh.data_ptr() + 4 * q), 2 * q, N, q, _stream())
        return gh


class _Elbo(torch.autograd.Function):
    """(loss, -mean lhood, mean KL(z0), KL(u)) of create_model.py:61-73 from the per-row terms, one launch."""

    @staticmethod
    def forward(ctx, lhood, klrow, kl_u, nobs):
        ctx.dims = (lhood.numel(), klrow.numel(), float(nobs), lhood.shape, klrow.shape, kl_u.shape)
        lhood, klrow, kl_u = _chk(lhood, 'lhood'), _chk(klrow, 'klrow'), _chk(kl_u, 'kl_u')
        out = _new((4,), lhood)
        _lib.call('gpode_elbo_fwd', _ptr(lhood), lhood.numel(), _ptr(klrow), klrow.numel(), _ptr(kl_u), ctypes.c_float(nobs), _ptr(out), _stream())
        return out

    @staticmethod
    def backward(ctx, gout):
        nl, nk, nobs, sl, sk, su = ctx.dims
        gl, gk, gu = _new(sl, gout), _new(sk, gout), _new(su, gout)
        _lib.call('gpode_elbo_bwd', _ptr(gout.contiguous()), nl, nk, ctypes.c_float(nobs), _ptr(gl), _ptr(gk), _ptr(gu), _stream())
        return gl, gk, gu, None


class _SigmoidLogLikParts(torch.autograd.Function):
    """Decoder logits -> partial row sums of the Bernoulli log-likelihood of X under sigmoid(logits) (gpode_sigmoid_loglik_fwd):
    the decoder's nn.Sigmoid (vae.py:84) and the summed log_prob (vae.py:136-153, create_model.py:49) in one pass over the logits."""

    @staticmethod
    def forward(ctx, X, a, rows):
        X, a = _chk(X, 'X'), _chk(a, 'logits')
        inner = a.numel() // rows
        ns = _lib.load().gpode_sigmoid_loglik_splits(rows, inner)
        z, part = torch.empty_like(a), _new((rows, ns), a)
        _lib.call('gpode_sigmoid_loglik_fwd', _ptr(X), _ptr(a), _ptr(z), _ptr(part), rows, inner, X.numel(), ns, _stream())
        ctx.save_for_backward(X, z)
        ctx.rows, ctx.inner = rows, inner
        ctx.mark_non_differentiable(z)
        ctx.set_materialize_grads(False)             # no zero-filled stand-in for the gradient of z (a fill of rows x inner floats per step)
        part._gpode_ll = (X, z)                      # for elbo_all(): its backward can then produce the logit gradients in the same launch
        return part, z

    @staticmethod
    def backward(ctx, gpart, _gz):
        X, z = ctx.saved_tensors
        ga = getattr(gpart, '_gpode_ga', None)       # already computed by _ElboAll.backward (gpode_elbo_all_bwd_ll)
        if ga is not None and ga.shape == z.shape:
            return None, ga, None
        # the ELBO hands every slice of a row the row's gradient: column 0 is the row gradient
        grow = gpart[:, 0].contiguous() if gpart.shape[1] > 1 else gpart.contiguous()
        ga = torch.empty_like(z)
        _lib.call('gpode_sigmoid_loglik_bwd', _ptr(X), _ptr(z), _ptr(grow), _ptr(ga), ctx.rows, ctx.inner, X.numel(), _stream())
        return None, ga, None


_ELBO_LL_FUSED = os.environ.get('GPODE_ELBO_LL_SEPARATE', '0') != '1'


class _ElboAll(torch.autograd.Function):
    """(loss, -mean lhood, mean KL(z0), KL(u)) of create_model.py:61-73 from the likelihood partial sums, the encoder's packed
    (mu | logvar) rows and the inducing posterior, one launch forward and one backward (gpode_elbo_all_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, lpart, hs, hv, Um, Us, rows, M, nobs, ll=None):
        ctx.ll = ll if (_ELBO_LL_FUSED and ll is not None and ll[1].numel() % ll[0].numel() == 0) else None
        lpart, hs, Um, Us = _chk(lpart, 'lpart'), _chk(hs, 'hs'), _chk(Um, 'Um'), _chk(Us, 'Us_sqrt.optvar')
        hv = _chk(hv, 'hv') if hv is not None else None
        N, q = hs.shape[0], hs.shape[1] // 2
        out = _new((4 + 256,), lpart)                # four results + the library's scratch
        _lib.call('gpode_elbo_all_fwd', _ptr(lpart), rows, lpart.numel(), _ptr(hs), _ptr(hv), N, q, M, Um.shape[1], _ptr(Um), _ptr(Us),
                  ctypes.c_float(nobs), _ptr(out), _stream())
        ctx.save_for_backward(hs, hv, Um, Us)
        ctx.dims = (rows, N, q, M, float(nobs), tuple(lpart.shape))
        ctx.set_materialize_grads(False)
        return out[0], out[1], out[2], out[3]

    @staticmethod
    def backward(ctx, g0, g1, g2, g3):
        hs, hv, Um, Us = ctx.saved_tensors
        rows, N, q, M, nobs, lshape = ctx.dims
        gs = [None if g is None else g.contiguous().float() for g in (g0, g1, g2, g3)]
        glrow = _new((rows,), hs)
        ghs, ghv = torch.empty_like(hs), (torch.empty_like(hv) if hv is not None else None)
        dUm, dUs = torch.empty_like(Um), torch.empty_like(Us)
        ga = None
        if ctx.ll is not None and ctx.needs_input_grad[0]:
            # the logits' gradient in the same launch (every likelihood row receives the same gradient): _SigmoidLogLikParts.backward
            # finds it on the row-gradient tensor it is handed and launches nothing
            X, z = ctx.ll
            ga = torch.empty_like(z)
            _lib.call('gpode_elbo_all_bwd_ll', *[_ptr(g) for g in gs], rows, _ptr(hs), _ptr(hv), N, q, M, Um.shape[1], _ptr(Um), _ptr(Us),
                      ctypes.c_float(nobs), _ptr(glrow), _ptr(ghs), _ptr(ghv), _ptr(dUm), _ptr(dUs), _ptr(X), _ptr(z), _ptr(ga), z.numel(),
                      X.numel(), _stream())
        else:
            _lib.call('gpode_elbo_all_bwd', *[_ptr(g) for g in gs], rows, _ptr(hs), _ptr(hv), N, q, M, Um.shape[1], _ptr(Um), _ptr(Us),
                      ctypes.c_float(nobs), _ptr(glrow), _ptr(ghs), _ptr(ghv), _ptr(dUm), _ptr(dUs), _stream())
        if (ctx.needs_input_grad[3] and ctx.needs_input_grad[4] and Um.is_leaf and Us.is_leaf and
                ops.defer_kl_grads((Um, Us), (dUm, dUs))):
            dUm = dUs = None                         # overlap mode: the flow's deferred backward adds its share to them in place
        # every slice of a likelihood row carries the row's gradient (a broadcast view: nothing is copied)
        glr = glrow.view(rows, 1).expand(lshape)
        if ga is not None:
            glr._gpode_ga = ga
        return glr, ghs, ghv, dUm, dUs, None, None, None, None


def sigmoid_loglik_parts(X, logits, rows):
    """-> (partial row sums (rows, nsplit) of the Bernoulli log-likelihood, z = sigmoid(logits))."""
    return _SigmoidLogLikParts.apply(X, logits, rows)


class _ElboAllKL(torch.autograd.Function):
    """_ElboAll on the KL partial sums of _ReparamKL instead of the packed (mu | logvar) rows (gpode_elbo_all_fwd_kl /
    gpode_elbo_all_bwd_ll_kl); its backward also produces the logits' gradient (ll = (X, z) of sigmoid_loglik_parts)."""

    @staticmethod
    def forward(ctx, lpart, kls, klv, Um, Us, rows, N, M, nobs, ll):
        lpart, kls, Um, Us = _chk(lpart, 'lpart'), _chk(kls, 'kl partial sums'), _chk(Um, 'Um'), _chk(Us, 'Us_sqrt.optvar')
        klv = _chk(klv, 'kl partial sums (v)') if klv is not None else None
        out = _new((4 + 256,), lpart)
        _lib.call('gpode_elbo_all_fwd_kl', _ptr(lpart), rows, lpart.numel(), _ptr(kls), kls.numel(), _ptr(klv), klv.numel() if klv is not None else 0,
                  N, M, Um.shape[1], _ptr(Um), _ptr(Us), ctypes.c_float(nobs), _ptr(out), _stream())
        ctx.save_for_backward(Um, Us)
        ctx.dims = (rows, N, M, float(nobs), tuple(lpart.shape), kls.numel(), klv.numel() if klv is not None else 0)
        ctx.ll = ll
        ctx.set_materialize_grads(False)
        return out[0], out[1], out[2], out[3]

    @staticmethod
    def backward(ctx, g0, g1, g2, g3):
        Um, Us = ctx.saved_tensors
        rows, N, M, nobs, lshape, nks, nkv = ctx.dims
        gs = [None if g is None else g.contiguous().float() for g in (g0, g1, g2, g3)]
        X, z = ctx.ll
        glrow, gkls, gklv = _new((rows,), Um), _new((nks,), Um), (_new((nkv,), Um) if nkv else None)
        dUm, dUs, ga = torch.empty_like(Um), torch.empty_like(Us), torch.empty_like(z)
        _lib.call('gpode_elbo_all_bwd_ll_kl', *[_ptr(g) for g in gs], rows, N, M, Um.shape[1], _ptr(Um), _ptr(Us), ctypes.c_float(nobs),
                  _ptr(glrow), _ptr(gkls), nks, _ptr(gklv), nkv, _ptr(dUm), _ptr(dUs), _ptr(X), _ptr(z), _ptr(ga), z.numel(), X.numel(), _stream())
        if (ctx.needs_input_grad[3] and ctx.needs_input_grad[4] and Um.is_leaf and Us.is_leaf and
                ops.defer_kl_grads((Um, Us), (dUm, dUs))):
            dUm = dUs = None
        glr = glrow.view(rows, 1).expand(lshape)
        glr._gpode_ga = ga
        return glr, gkls, gklv, dUm, dUs, None, None, None, None, None


def elbo_all(lpart, mu_s, logvar_s, mu_v, logvar_v, Um, Us_packed, M, nobs):
    """-> (loss, nll, kl_reg, kl_u), see _ElboAll."""
    ll = getattr(lpart, '_gpode_ll', None)
    hs0 = _packed_halves(mu_s, logvar_s)
    kls = getattr(hs0, '_gpode_klpart', None) if hs0 is not None else None
    klv = None
    if mu_v is not None:
        hv0 = _packed_halves(mu_v, logvar_v)
        klv = getattr(hv0, '_gpode_klpart', None) if hv0 is not None else None
    if (_ELBO_LL_FUSED and ll is not None and ll[1].numel() % ll[0].numel() == 0 and kls is not None and (mu_v is None or klv is not None)):
        return _ElboAllKL.apply(lpart, kls, klv, Um, Us_packed, lpart.shape[0], mu_s.shape[0], M, float(nobs), ll)
    hv = _pack(mu_v, logvar_v) if mu_v is not None else None
    return _ElboAll.apply(lpart, _pack(mu_s, logvar_s), hv, Um, Us_packed, lpart.shape[0], M, float(nobs), getattr(lpart, '_gpode_ll', None))


def _pack(mu, logvar):
    h = _packed_halves(mu, logvar)
    return h if h is not None else torch.cat((mu, logvar), dim=1)


_REPARAM_KL = os.environ.get('GPODE_REPARAM_KL_SEPARATE', '0') != '1'


def reparam(mu, logvar, eps):
    h = _packed_halves(mu, logvar)
    if h is not None and _REPARAM_KL and h.requires_grad:
        # (mu | logvar) are the halves of the encoder's fc output: the KL term's partial sums ride along for elbo_all()
        z, klpart = _ReparamKL.apply(h, eps)
        h._gpode_klpart = klpart
        return z
    return _Reparam.apply(_pack(mu, logvar), eps)


def normal_kl_rows(mu, logvar):
    return _NormalKL.apply(_pack(mu, logvar))


def elbo_terms(lhood_rows, kl_rows, kl_u, nobs):
    """-> tensor [loss, nll, kl_reg, kl_u] (views are taken by the caller)."""
    return _Elbo.apply(lhood_rows, kl_rows, kl_u, float(nobs))


def conv2d(x, w, b, stride, pad):
    return _Conv2d.apply(x, w, b, stride, pad)


def conv_transpose2d(x, w, b, stride, pad, out_pad=0, stats_for=None):
    """``stats_for``: the nn.BatchNorm2d (training mode) that follows -- its batch statistics are then summed by this convolution."""
    return _ConvT2d.apply(x, w, b, stride, pad, out_pad, stats_for)


def batch_norm_train(x, bn, relu):
    """nn.BatchNorm2d in training mode on the module's own parameters/buffers (updates running stats)."""
    # running statistics and the num_batches_tracked counter are updated by the kernel itself (one launch less per layer)
    return _BatchNormTrain.apply(x, bn.weight, bn.bias, bn.running_mean if bn.training else None,
                                 bn.running_var if bn.training else None, bn.momentum, bn.eps, relu,
                                 bn.num_batches_tracked if bn.training else None)


def bn_relu_conv_transpose2d(c, bn, w, b, stride, pad, out_pad=0, stats_for=None):
    """conv_transpose2d(relu(bn(c)), w, b) for a BatchNorm2d module in training mode, fused (see _BnReluConvT).  ``stats_for``: the
    BatchNorm2d that follows THIS convolution (see conv_transpose2d)."""
    return _BnReluConvT.apply(c, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum, bn.eps,
                              w, b, stride, pad, out_pad, bn, stats_for)


def batch_norm_eval(x, bn, relu):
    """nn.BatchNorm2d in evaluation mode on the module's running statistics."""
    return _BatchNormEval.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, relu)


def relu(x):
    return _Act.apply(x, 0)


def sigmoid(x):
    return _Act.apply(x, 1)


def linear(x, w, b):
    return _Linear.apply(x, w, b)


def linear_relu_in(x, w, b):
    """linear(relu(x), w, b); the ReLU is folded into the layer for wide fan-in (>= 128 inputs), applied separately otherwise."""
    if x.dim() == 2 and x.shape[1] >= 128 and x.shape[0] * w.shape[0] <= (1 << 22):
        return _LinearReluIn.apply(x, w, b)
    return _Linear.apply(relu(x), w, b)


def bernoulli_loglik(X, z):
    """log(z) X + log(1-z)(1-X), X broadcast over the leading copies of z (vae.py:136-153)."""
    return _LogLik.apply(X, z)


def bernoulli_loglik_rowsum(X, z, rows):
    return _LogLikRowSum.apply(X, z, rows)
