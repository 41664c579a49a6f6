"""Sampling FROM the energy model: `GFObjectPose.sample` on a posenet_mode='energy' agent.

The reference drives the same samplers with `self(data)` = PoseEnergyNet.forward(return_item='score'), i.e. the autograd
gradient of the inner-product energy <x, f(x)/sigma> with respect to the pose (networks/posenet.py:94-130,
networks/gf_algorithms/energynet.py:200-222) - NOT f/sigma.  On the device that score is one fused launch
(`gp_energy_score`, csrc/score_div.hip: forward trunk + vector-Jacobian product).  This is a secondary path (the runners sample
from the score model and only RANK with the energy model, evaluation_single.py:339-343), so the step arithmetic around the
launch stays host-driven:

  pc   one `gp_energy_score` launch per step + the corrector / predictor update as torch element-wise ops on the device
       (networks/gf_algorithms/samplers.py:102-160)
  ode  scipy's `solve_ivp(method='RK45')` on the host exactly as the reference runs it (samplers.py:163-227): the R x 9 state
       crosses PCIe once per function evaluation, the network evaluation is the fused launch.

The score agents' fused `gp_pc_step` / `gp_rk45_phase` kernels are not used here (they evaluate f/sigma).
"""
import math

import numpy as np
import torch

from .rotation import normalize_rotation
from .sde import EPS, SIGMA_MAX, SIGMA_MIN, ve_sde

_RATIO = SIGMA_MAX / SIGMA_MIN


def _score_at(net, cvec, K, x, t):
    """score of the energy model at one (uniform) diffusion time t (python float) -> [R,9] f32"""
    t32 = torch.full((1,), float(t), device=cvec.device, dtype=torch.float32)
    tvec = net.time_embed(t32)
    sigma = (SIGMA_MIN * _RATIO ** t32).contiguous()
    return net.energy_score(cvec, K, x.contiguous(), tvec[0], sigma)


def energy_pc_sample(net, cvec, K, centre, init_x, num_steps, z_langevin=None, z_predictor=None, snr=0.16, eps=EPS,
                     return_process=True):
    """cond_pc_sampler with the energy model's score.  net: ScoreNetHIP holding the ENERGY network's weights; cvec [B,768];
    centre [B,3]; init_x [B*K,9] f32.  Returns (xs [R,num_steps,9] or None, mean_x [R,9]) float32."""
    dev = cvec.device
    x = init_x.float().clone()
    R = x.shape[0]
    cen = centre.float().repeat_interleave(K, dim=0)
    time_steps = torch.linspace(1.0, eps, num_steps, device=dev)
    step_size = time_steps[0] - time_steps[1]
    tvec_all = net.time_embed(time_steps.contiguous())
    sigma_all = (SIGMA_MIN * _RATIO ** time_steps).contiguous()
    _, g_all = ve_sde(time_steps.reshape(-1, 1))
    noise_norm = math.sqrt(9)
    poses = []
    mean_x = None
    for i in range(num_steps):
        grad = net.energy_score(cvec, K, x.contiguous(), tvec_all[i], sigma_all[i:i + 1])
        grad_norm = torch.norm(grad, dim=-1).mean()            # batch-global coupling (samplers.py:130)
        lstep = 2 * (snr * noise_norm / grad_norm) ** 2
        z1 = torch.randn_like(x) if z_langevin is None else z_langevin[i]
        x = x + lstep * grad + torch.sqrt(2 * lstep) * z1
        x[:, :3] /= torch.norm(x[:, :3], dim=-1, keepdim=True)
        x[:, 3:6] /= torch.norm(x[:, 3:6], dim=-1, keepdim=True)
        g = g_all[i]
        drift = 0 - g ** 2 * grad                               # sign as written in the reference (:147), pre-corrector score
        mean_x = x + drift * step_size
        z2 = torch.randn_like(x) if z_predictor is None else z_predictor[i]
        x = mean_x + g * torch.sqrt(step_size) * z2
        x[:, :-3] = normalize_rotation(x[:, :-3])
        if return_process:
            poses.append(x.unsqueeze(0).clone())
    xs = None
    if return_process:
        xs = torch.cat(poses, dim=0)
        xs[:, :, -3:] += cen.unsqueeze(0)
        xs = xs.permute(1, 0, 2)
    mean_x = mean_x.clone()
    mean_x[:, -3:] += cen
    mean_x[:, :-3] = normalize_rotation(mean_x[:, :-3])
    return xs, mean_x


def energy_ode_sample(net, cvec, K, centre, init_x, T0, num_steps=None, eps=EPS, rtol=1e-5, atol=1e-5, denoise=True,
                      return_process=True, stats=None):
    """cond_ode_sampler with the energy model's score; the solver is scipy's RK45 on the host, as in the reference.
    Returns (xs [R,S,9] f64 or None, x [R,9] f64) on the device."""
    from scipy import integrate
    dev = cvec.device
    R = init_x.shape[0]
    cen = centre.double().repeat_interleave(K, dim=0)
    nfev = [0]

    def ode_func(t, y):
        x32 = torch.tensor(y.reshape(R, 9), dtype=torch.float32, device=dev)     # samplers.py:191
        score = _score_at(net, cvec, K, x32, t)
        g2 = (SIGMA_MIN * _RATIO ** float(t)) ** 2 * (2.0 * math.log(_RATIO))   # sde_coeff(torch.tensor(np.float64)) is f64 (:193)
        nfev[0] += 1
        return 0.0 - 0.5 * g2 * score.double().cpu().numpy().reshape(-1)

    t_eval = None if num_steps is None else np.linspace(T0, eps, num_steps)
    res = integrate.solve_ivp(ode_func, (T0, eps), init_x.detach().double().cpu().numpy().reshape(-1), rtol=rtol, atol=atol,
                              method="RK45", t_eval=t_eval)
    xs = torch.tensor(res.y, device=dev).T.reshape(-1, R, 9)
    x = torch.tensor(res.y[:, -1], device=dev).reshape(R, 9)
    if denoise:                                                                   # samplers.py:209-218
        t32 = torch.full((1,), eps, device=dev, dtype=torch.float32)
        _, g = ve_sde(t32)
        grad = _score_at(net, cvec, K, x.float(), eps)
        nfev[0] += 1
        x = x + (0 - g ** 2 * grad) * ((1 - eps) / (1000 if num_steps is None else num_steps))
    S = xs.shape[0]
    out = None
    if return_process:
        flat = xs.reshape(S * R, 9).clone()
        flat[:, :-3] = normalize_rotation(flat[:, :-3])
        out = flat.reshape(S, R, 9)
        out[:, :, -3:] += cen.unsqueeze(0)
        out = out.permute(1, 0, 2)
    x = x.clone()
    x[:, :-3] = normalize_rotation(x[:, :-3])
    x[:, -3:] += cen
    if stats is not None:
        stats["nfev"] = nfev[0]
    return out, x
