"""Is the replayed ELBO step bound by the HOST pushing the graph's packets?

For workloads cfg1 / cfg2: (a) host time to enqueue one replay (no synchronisation inside the loop) next to the device time per
step; (b) the same step captured TWICE and replayed alternately (two executable graphs: a launch of one never has to wait for its
own previous launch); (c) replays separated by a synchronisation (the host starts pushing only when the device is idle).
usage: python tools/probe_graph_cpu.py [cfg1 cfg2 ...]
"""
import os
import sys
import time

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench  # noqa: E402


def build(w, dev, ngraphs):
    from vae_gp_ode_amd import ops
    from vae_gp_ode_amd.graph import GraphedStep, device_generators
    from vae_gp_ode_amd.model.create_model import backward, compute_loss
    from vae_gp_ode_amd.optim import HipAdam
    model, X = bench.make_model_inputs(w, 0, dev, 0)
    Xd = X.to(dev)
    opt = HipAdam(model.parameters(), lr=1e-6, bucketed=False)

    def whole_step():
        opt.zero_grad()
        loss, *_ = compute_loss(model, Xd, 1)
        backward(loss)
        ops.join_side_stream()
        opt.step()
        return loss
    gs = [GraphedStep(whole_step, generators=device_generators(model), warmup=2 if i == 0 else 0) for i in range(ngraphs)]
    return gs


def timed(gs, steps, sync_each=False):
    for i in range(6):
        gs[i % len(gs)]()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        gs[i % len(gs)]()
        if sync_each:
            torch.cuda.synchronize()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / steps * 1e3, (t2 - t0) / steps * 1e3


def main():
    names = sys.argv[1:] or ['cfg1', 'cfg2']
    dev = torch.device('cuda:0')
    from vae_gp_ode_amd import ops
    ops.set_overlap(True)
    for n in names:
        w = bench.WORKLOADS[n]
        gs = build(w, dev, 2)
        for label, sel, sync in (('one graph', gs[:1], False), ('two graphs alternating', gs, False), ('one graph, sync each', gs[:1], True)):
            best = min((timed(sel, 200, sync) for _ in range(3)), key=lambda t: t[1])
            print('%s %-24s host enqueue %.3f ms/replay   total %.3f ms/step' % (n, label, best[0], best[1]), flush=True)


if __name__ == '__main__':
    main()
