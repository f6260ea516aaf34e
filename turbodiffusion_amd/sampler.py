"""The few-step rCM sampler loop of the reference (``inference/wan2.1_t2v_infer.py:111-140``;
I2V variant with the high/low-noise expert switch ``wan2.2_i2v_infer.py:173-213``), state in fp64 on
the GPU, the DiT step = ``turbodiffusion_amd.wan.WanModel.forward``."""
from __future__ import annotations

import math
from typing import Callable, List, Optional

import torch


def rcm_timesteps(num_steps: int = 4, sigma_max: float = 80.0, device="cpu") -> torch.Tensor:
    """TrigFlow times [atan(sigma_max), 1.5, 1.4, 1.0, 0] -> rectified-flow t = sin/(cos+sin)  (:111-122).
    Five fp64 numbers: computed on the HOST (the loop below needs them as host scalars — the expert switch compares
    t_cur with ``boundary`` — and a device tensor would cost one synchronising ``.item()`` per step)."""
    mid_t = [1.5, 1.4, 1.0][: num_steps - 1]
    t = torch.tensor([math.atan(sigma_max), *mid_t, 0], dtype=torch.float64)
    return (torch.sin(t) / (torch.cos(t) + torch.sin(t))).to(device)


def expert_schedule(num_steps: int = 4, sigma_max: float = 200.0, boundary: float = 0.9):
    """Which expert runs each step of the Wan2.2 loop (wan2.2_i2v_infer.py:190-197): 'high' while t_cur >= boundary,
    'low' from the first step with t_cur < boundary on.  sigma_max = 200, boundary 0.9 -> ['high', 'high', 'low', 'low']."""
    ts = rcm_timesteps(num_steps, sigma_max).tolist()
    out, switched = [], False
    for t_cur in ts[:-1]:
        switched = switched or t_cur < boundary
        out.append("low" if switched else "high")
    return out


def rcm_sample_iter(net: Callable, init_noise: torch.Tensor, crossattn_emb: torch.Tensor, num_steps: int = 4,
                    sigma_max: float = 80.0, generator: Optional[torch.Generator] = None,
                    noises: Optional[List[torch.Tensor]] = None, y: Optional[torch.Tensor] = None,
                    net_low: Optional[Callable] = None, boundary: float = 0.9, ode: bool = False,
                    dtype=torch.bfloat16):
    """The loop of ``rcm_sample`` as a generator: yields (step index, fp64 latent) after every sampler step, so that a
    caller can interleave several videos from ONE host thread (each resumed under its own stream).  Arguments as there.
    The yielded tensor is the sampler's STATE buffer, updated in place by the next step: clone it to keep it."""
    from . import kernels as K
    K.require_gpu(init_noise)                                 # (the CPU statement of this loop is oracle/wan_ref.rcm_sample)
    dev = init_noise.device
    t_host = rcm_timesteps(num_steps, sigma_max)             # fp64, host: no device sync anywhere in the loop
    t_steps = t_host.tolist()
    with torch.no_grad():
        x = (init_noise.to(torch.float64) * t_steps[0]).contiguous()
        x16 = x.to(dtype)
    kw = {} if y is None else {"y_B_C_T_H_W": y.to(dtype)}
    switched = False
    for i, (t_cur, t_next) in enumerate(zip(t_steps[:-1], t_steps[1:])):
        with torch.no_grad():
            switched = switched or (net_low is not None and t_cur < boundary)   # once low, stays low (:191-197)
            model = net_low if switched else net
            # (t_cur.float() * ones * 1000).to(dtype): the fp32 rounding of t_cur, times 1000 in fp64, cast  (:199)
            t_in = torch.full((x.size(0), 1), float(t_host[i].float()) * 1000.0, dtype=torch.float64, device=dev).to(dtype)
            v = model(x_B_C_T_H_W=x16, timesteps_B_T=t_in, crossattn_emb=crossattn_emb, **kw).float().contiguous()
            eps = None
            if not ode:
                eps = noises[i].to(dev).float().contiguous() if noises is not None else \
                    torch.randn(*x.shape, dtype=torch.float32, device=dev, generator=generator)
            # the update AND the next step's 16-bit network input in one pass over the state (td_rcm_step): the reference's
            # fp64 operator sequence operation for operation, its fp32 `t_next * randn` product included
            x16 = K.rcm_step_(x, v, eps, t_cur, t_next, dtype16=dtype)
        yield i, x


def rcm_sample(net: Callable, init_noise: torch.Tensor, crossattn_emb: torch.Tensor, num_steps: int = 4,
               sigma_max: float = 80.0, generator: Optional[torch.Generator] = None,
               noises: Optional[List[torch.Tensor]] = None, y: Optional[torch.Tensor] = None,
               net_low: Optional[Callable] = None, boundary: float = 0.9, ode: bool = False,
               dtype=torch.bfloat16, step_hook: Optional[Callable] = None) -> torch.Tensor:
    """x <- (1-t_next)(x - t_cur v) + t_next * N(0,1)   (SDE, :134-139) or x - (t_cur-t_next) v (ODE).

    ``noises`` (list of per-step N(0,1) tensors) overrides the generator — used by the parity tests so
    CPU oracle and GPU runs see identical noise.  ``net_low``/``boundary``: Wan2.2 expert switch
    (use ``net`` while t_cur >= boundary, ``net_low`` after: wan2.2_i2v_infer.py:191-197)."""
    x = None
    for i, x in rcm_sample_iter(net, init_noise, crossattn_emb, num_steps, sigma_max, generator, noises, y, net_low,
                                boundary, ode, dtype):
        if step_hook is not None:
            step_hook(i, x)
    return x.float()
