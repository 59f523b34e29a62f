"""Oracle: Euler-Maruyama SDE sampler with in-head CFG (test infrastructure only).

Restates /root/reference/modeling/vision_head/sampling_x.py:
  time_shift_func          :3-4
  get_score_from_velocity  :6-13
  get_velocity_from_cfg    :16-20
  euler_step               :24-29
  euler_maruyama_step      :33-41
  euler_maruyama           :44-97

Every elementwise op is an individually rounded fp32 torch op in the reference's
evaluation order, so on identical inputs this is bit-identical to the reference
(pinned by tests/golden/sampler_*.npz).
"""
from __future__ import annotations

from typing import Callable, Iterator, Sequence

import torch

F32 = torch.float32


def shifted_times(n_steps: int, last_step: float = 0.05, time_shift: float = 1.0) -> torch.Tensor:
    """t_all after the (identity-for-shift-1 but not bitwise) time shift. sampling_x.py:3-4,62-63."""
    t = torch.linspace(0, 1 - last_step, n_steps + 1, dtype=F32)
    inv = 1 / time_shift
    return inv / (inv + (1 / t - 1) ** 1.0)


def step_table(n_steps: int, last_step: float = 0.05, time_shift: float = 1.0):
    """The data-independent scalars of every sampling step, as fp32 0-dim tensors.

    Returns (ts, dts): ts[i] is the running time `t` *before* SDE step i (sampling_x.py:65-67,83:
    starts at 0.0 and accumulates ``t += dt[i]`` in fp32), dts[i] = t_all[i+1]-t_all[i] (:64).
    """
    t_all = shifted_times(n_steps, last_step, time_shift)
    dts = t_all[1:] - t_all[:-1]
    ts = []
    t = torch.tensor(0.0, dtype=F32)
    for i in range(n_steps):
        ts.append(t.clone())
        t = t + dts[i]
    return ts, [dts[i] for i in range(n_steps)]


def cfg_mix(v: torch.Tensor, cfg: float, cfg_mult: int) -> torch.Tensor:
    """sampling_x.py:16-20."""
    if cfg_mult == 2:
        v_c, v_u = torch.chunk(v, 2, dim=0)
        v = v_u + cfg * (v_c - v_u)
    return v


def velocity_from_xhat(xhat: torch.Tensor, combined: torch.Tensor, t_rows: torch.Tensor) -> torch.Tensor:
    """v = (x_hat - x) / clamp_min(1 - t, 0.05).  sampling_x.py:77-80,91-94."""
    shape = [-1] + [1] * (xhat.dim() - 1)
    return (xhat - combined) / (1 - t_rows.view(*shape)).clamp_min(0.05)


def sde_step(x, v, t, dt, cfg: float, cfg_mult: int, eps):
    """One Euler-Maruyama update. sampling_x.py:33-41 with :6-13 inlined in evaluation order."""
    v = v.to(F32)
    v = cfg_mix(v, cfg, cfg_mult)
    sigma_t = 1 - t
    rar = t / 1                      # alpha_t / d_alpha_t
    var = sigma_t ** 2 - rar * -1 * sigma_t
    score = (rar * v - x) / var
    drift = v + (1 - t) * score
    noise_scale = (2.0 * (1.0 - t) * dt) ** 0.5
    return x + drift * dt + noise_scale * eps


def ode_step(x, v, dt: float, cfg: float, cfg_mult: int):
    """Final noiseless Euler step. sampling_x.py:24-29."""
    v = cfg_mix(v.to(F32), cfg, cfg_mult)
    return x + v * dt


def euler_maruyama(
    input_dim: int,
    forward_fn: Callable[[torch.Tensor, torch.Tensor, torch.Tensor], torch.Tensor],
    c: torch.Tensor,
    cfg: float,
    num_sampling_steps: int,
    noise: Iterator[torch.Tensor] | Sequence[torch.Tensor],
    last_step_size: float = 0.05,
    time_shift: float = 1.0,
    trace: list | None = None,
) -> torch.Tensor:
    """sampling_x.py:44-97.  ``noise`` yields, in the reference's RNG call order,
    the initial latent ``[B,P,C]`` then one ``[B,P,C]`` tensor per SDE step."""
    noise = iter(noise)
    cfg_mult = 2 if cfg > 1.0 else 1
    x = next(noise).to(device=c.device, dtype=F32)
    assert x.shape[0] == c.shape[0] // cfg_mult and x.shape[-1] == input_dim
    ts, dts = step_table(num_sampling_steps, last_step_size, time_shift)
    t_rows = torch.zeros(c.shape[0], dtype=F32, device=c.device)     # (device-aware: tests/test_gpu_autocast.py runs this loop under the device's autocast)
    for i in range(num_sampling_steps):
        t_rows[:] = ts[i]
        combined = torch.cat([x] * cfg_mult, dim=0)
        xhat = forward_fn(combined, t_rows, c)
        v = velocity_from_xhat(xhat, combined, t_rows)
        x = sde_step(x, v, ts[i], dts[i], cfg, cfg_mult, next(noise).to(device=c.device, dtype=F32))
        if trace is not None:
            trace.append(x.clone())
    combined = torch.cat([x] * cfg_mult, dim=0)
    t_rows[:] = 1 - last_step_size
    xhat = forward_fn(combined, t_rows, c)
    v = velocity_from_xhat(xhat, combined, t_rows)
    x = ode_step(x, v, last_step_size, cfg, cfg_mult)
    if trace is not None:
        trace.append(x.clone())
    return torch.cat([x] * cfg_mult, dim=0)
