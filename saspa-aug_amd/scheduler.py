"""DDIM scheduler (host-side state only; the update itself is the fused HIP kernel
saspa_cfg_ddim_step).  Mirrors `DDIMScheduler.from_config(pipe.scheduler.config)` as built
at run_aug/run_aug.py:221 from the SD-1.5 scheduler config: scaled_linear betas
0.00085..0.012 over 1000 train steps, steps_offset=1, set_alpha_to_one=False,
clip_sample=False, epsilon prediction, "leading" timestep spacing, eta=0."""
import numpy as np
import torch

SD15_SCHEDULER_CONFIG = dict(
    num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
    steps_offset=1, set_alpha_to_one=False, clip_sample=False, prediction_type="epsilon",
    timestep_spacing="leading",
)


class DDIMScheduler:
    def __init__(self, **config):
        self.config = dict(SD15_SCHEDULER_CONFIG)
        self.config.update(config)
        c = self.config
        if c["beta_schedule"] != "scaled_linear" or c["prediction_type"] != "epsilon" or c["clip_sample"]:
            raise NotImplementedError("only the SD-1.5 DDIM configuration is implemented")
        n = c["num_train_timesteps"]
        betas = torch.linspace(c["beta_start"] ** 0.5, c["beta_end"] ** 0.5, n, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if c["set_alpha_to_one"] else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.timesteps = None
        self.num_inference_steps = None

    @classmethod
    def from_config(cls, config):
        return cls(**dict(config))

    def set_timesteps(self, num_inference_steps):
        c = self.config
        self.num_inference_steps = num_inference_steps
        n = c["num_train_timesteps"]
        if c["timestep_spacing"] == "leading":
            ratio = n // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + c["steps_offset"]
        elif c["timestep_spacing"] == "trailing":
            ts = np.round(np.arange(n, 0, -n / num_inference_steps)).astype(np.int64) - 1
        else:
            raise NotImplementedError(c["timestep_spacing"])
        self.timesteps = ts
        return ts

    def step_coefficients(self, t):
        """(sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)) in fp32, eta = 0."""
        n = self.config["num_train_timesteps"]
        prev_t = int(t) - n // self.num_inference_steps
        a_t = self.alphas_cumprod[int(t)]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return (float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(a_prev ** 0.5), float((1 - a_prev) ** 0.5))
