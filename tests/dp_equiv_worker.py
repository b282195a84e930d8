"""Worker of tests/test_gpu_sync_bn.py: `steps` full training steps (encoder, GP draw, rk4 rollout, decoder, ELBO, backward,
gradient all-reduce, a normalised gradient step) on this rank's shard of a FIXED global minibatch, with every random input derived from the global
batch (encoder noise eps[shard], one GP draw shared by all ranks).  Run with WORLD_SIZE = 1 it is the reference's single-process
step on the whole minibatch; with WORLD_SIZE = N over gloo (all ranks on one card) the N-rank step.  Rank 0 saves gradients of the
first step and the parameters / BatchNorm buffers after the last one."""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, n_global, steps, sync_bn = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == '1'
    kernel = sys.argv[5] if len(sys.argv) > 5 else 'RBF'
    order = 2 if kernel.endswith('2') else 1         # 'RBF2': second-order model (position + velocity encoder, vae.py:14-19)
    kernel = kernel.rstrip('2')
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    from vae_gp_ode_amd import vae_ops
    from vae_gp_ode_amd.model.core.initialization import initialize_and_fix_kernel_parameters
    from vae_gp_ode_amd.model.create_model import build_model, compute_loss
    from vae_gp_ode_amd.model.misc.torch_utils import seed_everything
    from vae_gp_ode_amd.optim import HipAdam
    from vae_gp_ode_amd.parallel import BatchNormSync, GradAllReduce, shard_bounds
    torch.cuda.set_device(0)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('gloo')
    q, M, S, T = (6 if order == 1 else 3), 16, 32, 6
    args = types.SimpleNamespace(D_in=q * order, D_out=q, num_inducing=M, num_features=S, dimwise=True, q_diag=False, device='cuda', kernel=kernel,
                                 ode=order, solver='rk4', use_adjoint=False, frames=5, n_filt=8, latent_dim=q, Ndata=360, dt=0.1)
    seed_everything(11)
    m = build_model(args).cuda()
    initialize_and_fix_kernel_parameters(m, 2.0, 1.0)
    g = torch.Generator().manual_seed(12)
    X = torch.rand(n_global, T, 1, 28, 28, generator=g)
    lo, hi = shard_bounds(n_global, rank, world)
    # lr 1e-4: with the reference's 1e-3 = the initial diag(Us_sqrt), Adam's first step puts diagonal entries on exactly 0 (log 0 in KL(u))
    opt = HipAdam(m.parameters(), lr=1e-4, bucketed='gather')
    sync = GradAllReduce(opt.flat_grads, dist, weight=(hi - lo) / n_global) if dist is not None else None
    if dist is not None and sync_bn:
        bs = BatchNormSync(dist, shares=[b - a for a, b in (shard_bounds(n_global, r, world) for r in range(world))])
        vae_ops.set_bn_sync(bs)
    gp = m.flow.odefunc.diffeq
    first_grads = None
    for it in range(steps):
        nz = dict(eps_u=torch.randn(M, q, generator=g), rff_w=torch.randn(S if kernel == 'RBF' else 2 * S, q, generator=g),
                  rff_eps=torch.randn(q * order, S, q, generator=g), rff_u=torch.rand(1, S, q, generator=g))
        eps = torch.randn(n_global, q, generator=g)
        gp.set_noise({k: v.cuda() for k, v in nz.items()})
        m.vae.encoder.next_eps = eps[lo:hi].cuda()
        if order == 2:
            m.vae.encoder_v.next_eps = torch.randn(n_global, q, generator=g)[lo:hi].cuda()
        opt.zero_grad()
        loss, *_ = compute_loss(m, X[lo:hi].cuda(), 1)
        loss.backward()
        if sync is not None:
            sync.all_reduce_grads()
            flat = opt.flat_grads.flat
        else:
            opt.gather_grads()
            flat = opt.flat_grads.flat
        if first_grads is None:
            first_grads = {k: flat[o:o + p.numel()].view_as(p).detach().cpu().clone()
                           for (k, p), o in zip([(k, p) for k, p in m.named_parameters() if p.requires_grad], opt.flat_grads.offsets)}
        # A plain gradient step (one global scale) instead of Adam: Adam divides every entry by its own magnitude, so an entry whose
        # gradient is round-off noise (the zero-initialised off-diagonals of Us_sqrt, the biases in front of a BatchNorm) moves by
        # +-lr whatever its sign -- that amplification is the optimiser's, not the data-parallel step's (HipAdam: test_gpu_optim.py).
        with torch.no_grad():
            scale = 1e-3 / float(flat.abs().max())
            for p, o in zip(opt.flat_grads.params, opt.flat_grads.offsets):
                p.add_(flat[o:o + p.numel()].view_as(p), alpha=-scale * float(p.abs().max()))
    torch.cuda.synchronize()
    if rank == 0:
        torch.save(dict(grads=first_grads, state={k: v.detach().cpu() for k, v in m.state_dict().items()}), out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
