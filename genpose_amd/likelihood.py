"""cond_ode_likelihood (networks/gf_algorithms/samplers.py:22-99) - SURVEY §8f row 3.

The instantaneous change-of-variables ODE  d[x, logp]/dt = [-g(t)^2/2 * score(x, t), -g(t)^2/2 * div_x score(x, t)]  integrated from
eps to 1 with Dormand-Prince 5(4) under scipy's step controller (`integrate.solve_ivp(method='RK45')`, rtol = atol = 1e-5, one error
norm over the whole [x; logp] vector), the divergence by the Skilling-Hutchinson estimator with ONE fixed probe eps ~ prior.  The
reference runs the solver on the host: two network evaluations (one of them under autograd) and four PCIe crossings per function
call.  Here the whole solve is resident on the device - the RK45 driver of the samplers (csrc/rk45.hip, model 2) with a ten-component
state per row, score and divergence from one fused forward + vector-Jacobian pass per stage (csrc/score_bwd.h) - and the host reads
one status word per replayed chunk of attempts.
"""
import math

import torch

from .samplers import ODESampler
from .sde import SIGMA_MAX


def global_prior_likelihood(z, sigma_max):
    """log N(z; 0, sigma_max^2 I) per row (samplers.py:13-19)."""
    n = z.shape[1]
    return -n / 2.0 * math.log(2 * math.pi * sigma_max ** 2) - torch.sum(z ** 2, dim=-1) / (2 * sigma_max ** 2)


def cond_ode_likelihood(net, cvec, k, x, epsilon, eps=1e-5, rtol=1e-5, atol=1e-5, stats=None, solver=None):
    """net: ScoreNetHIP; cvec [B,768] (gp_cloud_embed); x [B*k,9] poses whose likelihood is wanted; epsilon [B*k,9] the fixed
    Hutchinson probe (the reference draws it from the prior, samplers.py:39).  Returns (z [R,9] f64, log-likelihood in bits [R] f64)
    on the device.  solver: an ODESampler(model='likelihood') of the right shape to reuse (buffers, captured attempts)."""
    B = cvec.shape[0]
    if solver is None:
        solver = ODESampler(net, B, k, cvec.device, model="likelihood")
    z, delta_logp = solver.run_likelihood(cvec, x.float().contiguous(), epsilon.to(cvec.device).float().contiguous(), eps=eps, rtol=rtol, atol=atol)
    nll = (global_prior_likelihood(z, SIGMA_MAX) + delta_logp) / math.log(2)
    if stats is not None:
        stats["nfev"] = int(solver.last_stats["nfev"])
        stats["attempts"] = int(solver.last_stats["n_attempts"])
    return z, nll
