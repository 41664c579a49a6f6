"""VE-SDE coefficient functions (host side), mirroring networks/gf_algorithms/sde.py:15-28,90-97.
Only the VE SDE is on the hot path (default --sde_mode ve, configs/config.py:34)."""
import functools

import numpy as np
import torch

SIGMA_MIN, SIGMA_MAX, EPS, T_MAX = 0.01, 50.0, 1e-5, 1.0


def ve_marginal_prob(x, t, sigma_min=SIGMA_MIN, sigma_max=SIGMA_MAX):
    std = sigma_min * (sigma_max / sigma_min) ** t
    return x, std


def ve_sde(t, sigma_min=SIGMA_MIN, sigma_max=SIGMA_MAX):
    sigma = sigma_min * (sigma_max / sigma_min) ** t
    drift_coeff = torch.tensor(0)
    diffusion_coeff = sigma * torch.sqrt(torch.tensor(2 * (np.log(sigma_max) - np.log(sigma_min)), device=t.device))
    return drift_coeff, diffusion_coeff


def ve_prior(shape, sigma_min=SIGMA_MIN, sigma_max=SIGMA_MAX, T=1.0):
    """Drawn on the CPU generator exactly like the reference (sde.py:26-28), then moved by the caller."""
    _, sigma_max_prior = ve_marginal_prob(None, T, sigma_min=sigma_min, sigma_max=sigma_max)
    return torch.randn(*shape) * sigma_max_prior


def init_sde(sde_mode):
    if sde_mode != "ve":
        raise NotImplementedError(f"sde_mode '{sde_mode}': only the VE SDE is implemented on the MI355X hot path")
    prior_fn = functools.partial(ve_prior, sigma_min=SIGMA_MIN, sigma_max=SIGMA_MAX)
    marginal_prob_fn = functools.partial(ve_marginal_prob, sigma_min=SIGMA_MIN, sigma_max=SIGMA_MAX)
    sde_fn = functools.partial(ve_sde, sigma_min=SIGMA_MIN, sigma_max=SIGMA_MAX)
    return prior_fn, marginal_prob_fn, sde_fn, EPS, T_MAX
