"""ODE right-hand side wrapper and fixed-grid flow -- operator API of experiments/model/core/flow.py.

The reference hands ``ODEfunc`` to torchdiffeq, which calls it 1 (euler) or 4 (rk4) times per step
from Python.  Here ``Flow.forward`` is ONE persistent HIP kernel per MC draw (csrc/gp_forward.hip):
each wavefront integrates one trajectory over the whole grid.  Fixed-grid methods only ('euler',
'rk4' = torchdiffeq's 3/8 rule, 'midpoint'); adaptive solvers are out of scope.
"""
import torch
import torch.nn as nn

from ... import ops

EVALS_PER_STEP = {'euler': 1, 'rk4': 4, 'midpoint': 2}


class ODEfunc(nn.Module):
    def __init__(self, diffeq, order):
        super().__init__()
        self.diffeq = diffeq
        self.order = order
        self.register_buffer('_num_evals', torch.tensor(0.))
        self._host_evals = None                      # count of the last fused solve, not yet written to the buffer

    def before_odeint(self, rebuild_cache):
        self._host_evals = None
        self._num_evals.fill_(0)
        if rebuild_cache:
            self.diffeq.build_cache()

    def _set_evals(self, n):
        """The fused rollout knows its evaluation count on the host: the ``_num_evals`` buffer (a state_dict entry of the
        reference, flow.py:14) is brought up to date when somebody looks -- not by a fill and an add on the device in every step."""
        self._host_evals = float(n)

    def _flush_evals(self):
        if self._host_evals is not None:
            self._num_evals.fill_(self._host_evals)
            self._host_evals = None

    def num_evals(self):
        return self._host_evals if self._host_evals is not None else self._num_evals.item()

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self._flush_evals()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def first_order(self, sv):
        return self.diffeq(sv)

    def second_order(self, sv):
        q = sv.shape[1] // 2
        return torch.cat([sv[:, q:], self.diffeq(sv)], 1)

    def forward(self, t, sv):
        """One RHS evaluation (flow.py:40-45); autonomous, ``t`` ignored."""
        self._flush_evals()
        self._num_evals += 1
        return self.first_order(sv) if self.order == 1 else self.second_order(sv)


class Flow(nn.Module):
    def __init__(self, diffeq, order=2, solver='dopri5', atol=1e-6, rtol=1e-6, use_adjoint=False):
        super().__init__()
        self.odefunc = ODEfunc(diffeq, order)
        self.solver = solver
        self.atol, self.rtol = atol, rtol  # ignored by fixed-grid methods (as in torchdiffeq)
        self.use_adjoint = use_adjoint     # same forward; gradients are discretise-then-optimise either way

    def forward(self, z0, ts, draws=None):
        """z0 (N,D), ts (T,) -> zt (N,T,D) for a fresh function draw (flow.py:68-86).  ``draws`` = L: L fresh draws integrated in
        one pass -> (L,N,T,D), the stack ODEGPVAE.sample_trajectories builds from L calls (odegpvae.py:41-44); ``_num_evals`` ends
        at the count of ONE solve, as it does after the reference's last call."""
        if self.solver not in EVALS_PER_STEP:
            raise ValueError("solver '%s': this build integrates on the fixed grid with 'euler', 'midpoint' or 'rk4' only" % self.solver)
        gp = self.odefunc.diffeq
        zt = ops.flow(gp, z0, ts, self.odefunc.order, self.solver, draws)
        self.odefunc._set_evals(EVALS_PER_STEP[self.solver] * (ts.shape[0] - 1))
        return zt

    def num_evals(self):
        return self.odefunc.num_evals()

    def kl(self):
        return self.odefunc.diffeq.kl()
