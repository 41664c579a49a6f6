"""cond_ode_likelihood (networks/gf_algorithms/samplers.py:22-99) - SURVEY §8f row 3.

The instantaneous change-of-variables ODE  d[x, logp]/dt = [-g(t)^2/2 * score(x, t), -g(t)^2/2 * div_x score(x, t)]  integrated from
eps to 1 with scipy's RK45 exactly as the reference does (`integrate.solve_ivp`, rtol = atol = 1e-5, one error norm over the whole
[x; logp] vector), the divergence by the Skilling-Hutchinson estimator with ONE fixed probe eps ~ prior.  The reference pays
two network evaluations (one of them under autograd) and four PCIe crossings per function call; here a function call is one
fused launch (gp_score_div) plus the time embedding, and only the R x 10 state vector crosses PCIe.  This is not on the
hot path of the benchmarks: the solver loop stays scipy's on the host, which also keeps its step control bit-identical.
"""
import math

import numpy as np
import torch
from scipy import integrate

from .sde import SIGMA_MAX, SIGMA_MIN


def global_prior_likelihood(z, sigma_max):
    """log N(z; 0, sigma_max^2 I) per row (samplers.py:13-19)."""
    n = z.shape[1]
    return -n / 2.0 * math.log(2 * math.pi * sigma_max ** 2) - torch.sum(z ** 2, dim=-1) / (2 * sigma_max ** 2)


def cond_ode_likelihood(net, cvec, k, x, epsilon, eps=1e-5, rtol=1e-5, atol=1e-5, stats=None):
    """net: ScoreNetHIP; cvec [B,768] (gp_cloud_embed); x [B*k,9] poses whose likelihood is wanted; epsilon [B*k,9] the fixed
    Hutchinson probe (the reference draws it from the prior, samplers.py:39).  Returns (z [R,9] f64, log-likelihood in bits [R] f64)
    on the device."""
    dev = cvec.device
    R = x.shape[0]
    epsilon = epsilon.to(dev).float().contiguous()
    init = np.concatenate([x.detach().double().cpu().numpy().reshape(-1), np.zeros(R)])
    ratio = SIGMA_MAX / SIGMA_MIN
    nfev = [0]

    def ode_func(t, inp):
        xt = torch.tensor(inp[:-R].reshape(R, 9), dtype=torch.float32, device=dev)   # samplers.py:80
        t32 = torch.ones(1, device=dev) * t                                           # the network's time input is f32 (:81)
        tvec = net.time_embed(t32.float().contiguous())
        sigma32 = (SIGMA_MIN * ratio ** t32.float()).contiguous()                      # marginal_prob on the f32 time tensor
        score, div = net.score_and_divergence(cvec, k, xt, epsilon, tvec[0], sigma32)
        g2 = (SIGMA_MIN * ratio ** float(t)) ** 2 * (2.0 * math.log(ratio))           # sde_coeff(torch.tensor(np.float64)): f64 (:82)
        nfev[0] += 1
        x_grad = 0.0 - 0.5 * g2 * score.double().cpu().numpy().reshape(-1)
        logp_grad = 0.0 - 0.5 * g2 * div.double().cpu().numpy().reshape(-1)
        return np.concatenate([x_grad, logp_grad])

    res = integrate.solve_ivp(ode_func, (eps, 1.0), init, rtol=rtol, atol=atol, method="RK45")
    zp = torch.tensor(res.y[:, -1], device=dev)
    z = zp[:-R].reshape(R, 9)
    delta_logp = zp[-R:]
    nll = (global_prior_likelihood(z, SIGMA_MAX) + delta_logp) / math.log(2)
    if stats is not None:
        stats["nfev"] = nfev[0]
    return z, nll
