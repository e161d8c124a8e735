"""DPM-Solver++ (2M, midpoint) schedule tables for the fused sampling kernel.

Host-side mirror of the part of the reference's `DPMSolverMultistepScheduler` the inference path uses
(vibevoice/schedule/dpm_solver.py: cosine betas :53-54,78-83; ctor :203-295; set_timesteps :321-423;
_sigma_to_alpha_sigma_t :483-487; v-prediction convert_model_output :581-584; first-order update :669-677;
second-order midpoint update :738-764; order selection in step :976-1003).  Only scalars are computed here
(once per `set_timesteps`), with the same fp32 0-dim tensor arithmetic as the reference; the per-element
update itself runs on the GPU (`vv_dpm_step`).
"""
from __future__ import annotations

import math
from typing import List, Optional

import numpy as np
import torch


class DPMSolverMultistepScheduler:
    """API-compatible subset: ctor kwargs, `.config`, `.from_config`, `.set_timesteps`, `.timesteps`, `.sigmas`."""

    def __init__(self, num_train_timesteps: int = 1000, beta_schedule: str = "cosine", prediction_type: str = "v_prediction",
                 solver_order: int = 2, algorithm_type: str = "dpmsolver++", solver_type: str = "midpoint",
                 final_sigmas_type: str = "zero", timestep_spacing: str = "linspace", **unused):
        if beta_schedule not in ("cosine", "squaredcos_cap_v2"):
            raise NotImplementedError(f"beta_schedule {beta_schedule!r}: only the cosine schedule of the shipped configs is built")
        if prediction_type != "v_prediction" or algorithm_type not in ("dpmsolver++", "sde-dpmsolver++") or solver_type != "midpoint" \
                or solver_order != 2 or final_sigmas_type != "zero" or timestep_spacing != "linspace":
            raise NotImplementedError("only dpmsolver++ (every shipped model) and sde-dpmsolver++ (main.py:543-548) / order 2 / midpoint / "
                                      "v_prediction / linspace / final sigma 0 are built")
        self.config = dict(num_train_timesteps=num_train_timesteps, beta_schedule=beta_schedule, prediction_type=prediction_type,
                           solver_order=solver_order, algorithm_type=algorithm_type, solver_type=solver_type,
                           final_sigmas_type=final_sigmas_type, timestep_spacing=timestep_spacing)

        def alpha_bar(t):
            return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        n = num_train_timesteps
        betas = torch.tensor([min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), 0.999) for i in range(n)], dtype=torch.float32)
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.arange(n - 1, -1, -1, dtype=torch.int64)
        self.sigmas = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        self.coefs: List[dict] = []

    @classmethod
    def from_config(cls, config, **overrides):
        kw = dict(config)
        kw.update(overrides)
        return cls(**kw)

    def set_timesteps(self, num_inference_steps: int, device=None):
        n_train = self.alphas_cumprod.shape[0]
        ts = np.linspace(0, n_train - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        sig = np.concatenate([sig, [0]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sig)
        self.timesteps = torch.from_numpy(ts)
        self.num_inference_steps = len(ts)
        self.coefs = self._coefficients(self.sigmas, self.config["algorithm_type"])

    @staticmethod
    def _alpha_sigma(sigma):
        alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
        return alpha_t, sigma * alpha_t

    @classmethod
    def _coefficients(cls, sigmas: torch.Tensor, algorithm: str = "dpmsolver++") -> List[dict]:
        n = sigmas.shape[0] - 1
        out = []
        for i in range(n):
            a_s0, s_s0 = cls._alpha_sigma(sigmas[i])
            a_t, s_t = cls._alpha_sigma(sigmas[i + 1])
            lam_t = torch.log(a_t) - torch.log(s_t)
            lam_s0 = torch.log(a_s0) - torch.log(s_s0)
            h = lam_t - lam_s0
            c = dict(alpha_s=float(a_s0), sigma_s=float(s_s0), cx=float(s_t / s_s0),
                     cd=float(a_t * (torch.exp(-h) - 1.0)), rinv=0.0, order=1, cn=0.0)
            if algorithm == "sde-dpmsolver++":       # dpm_solver.py:680-686, :785-793: same update shape plus a variance-noise term
                c["cx"] = float(s_t / s_s0 * torch.exp(-h))
                c["cd"] = float(-(a_t * (1 - torch.exp(-2.0 * h))))
                c["cn"] = float(s_t * torch.sqrt(1.0 - torch.exp(-2 * h)))
            if 0 < i < n - 1:            # first step and (final sigma 0) last step are first order
                a_s1, s_s1 = cls._alpha_sigma(sigmas[i - 1])
                lam_s1 = torch.log(a_s1) - torch.log(s_s1)
                r0 = (lam_s0 - lam_s1) / h
                c["order"], c["rinv"] = 2, float(1.0 / r0)
            out.append(c)
        return out


def timestep_sinusoid(timesteps, dim: int = 256, max_period: float = 10000.0, bf16_quirk: bool = False) -> torch.Tensor:
    """TimestepEmbedder.timestep_embedding (modular_vibevoice_diffusion_head.py:66-88) on the host, fp32.
    bf16_quirk reproduces the reference's bf16 run, where `t` is cast to the activation dtype before the
    sinusoid (modeling_vibevoice_inference.py:703) and the embedding is cast back to it (:88)."""
    t = torch.as_tensor(np.asarray(timesteps), dtype=torch.float32)
    if bf16_quirk:
        t = t.to(torch.bfloat16).float()
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None] * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if bf16_quirk:
        emb = emb.to(torch.bfloat16).float()
    return emb
