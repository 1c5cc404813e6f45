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


# stabilityai/sdxl-turbo scheduler/scheduler_config.json (EulerAncestralDiscreteScheduler) as seen by
# `DDIMScheduler.from_config` at run_aug/run_aug.py:228: the keys DDIM understands are kept -- "trailing" spacing,
# the SD beta schedule, and the SDXL repos' leftover clip_sample=false / set_alpha_to_one=false.  Recalled (no
# network here): parity unpinned.
SDXL_TURBO_SCHEDULER_CONFIG = dict(
    num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
    steps_offset=1, set_alpha_to_one=False, clip_sample=False, prediction_type="epsilon",
    timestep_spacing="trailing",
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


class PNDMScheduler:
    """`PNDMScheduler` with `skip_prk_steps=True` (pure PLMS), the scheduler the SD-1.5-derived BLIP-Diffusion
    pipeline ships and the reference keeps for it (run_aug/run_aug.py:217: only non-BLIP pipelines are switched to
    DDIM).  Host-side state only; `plan()` turns a step count into the list of network evaluations with the
    coefficients of the fused update kernel (saspa_cfg_plms_step).  [upstream] diffusers 0.32.2 scheduling_pndm.py,
    recalled: N inference steps cost N+1 evaluations (the second timestep is visited twice)."""

    def __init__(self, **config):
        self.config = dict(SD15_SCHEDULER_CONFIG, skip_prk_steps=True)
        self.config.update(config)
        c = self.config
        if c["beta_schedule"] != "scaled_linear" or c["prediction_type"] != "epsilon" or not c["skip_prk_steps"]:
            raise NotImplementedError("only the SD-1.5 PNDM configuration (PLMS, epsilon) is implemented")
        if c["timestep_spacing"] != "leading":
            raise NotImplementedError(c["timestep_spacing"])
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
        ratio = c["num_train_timesteps"] // num_inference_steps
        base = (np.arange(0, num_inference_steps) * ratio).round().astype(np.int64) + c["steps_offset"]
        plms = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy()
        self.timesteps = plms
        return plms

    def _coefficients(self, timestep, prev_timestep):
        a_t = float(self.alphas_cumprod[timestep])
        a_p = float(self.alphas_cumprod[prev_timestep]) if prev_timestep >= 0 else float(self.final_alpha_cumprod)
        sample_coeff = (a_p / a_t) ** 0.5
        denom = a_t * (1 - a_p) ** 0.5 + (a_t * (1 - a_t) * a_p) ** 0.5
        return sample_coeff, -(a_p - a_t) / denom

    def plan(self, num_inference_steps=None):
        """[(t_network, dict(store_slot, w_cur, w_hist[4], coef_sample, coef_model, save_sample, use_saved))].
        `ets` of the reference is a ring of 4 device slots; slot bookkeeping lives here."""
        if num_inference_steps is not None:
            self.set_timesteps(num_inference_steps)
        ratio = self.config["num_train_timesteps"] // self.num_inference_steps
        out, ets = [], []            # ets: slot indices, oldest first
        next_slot = 0
        for counter, t in enumerate(int(v) for v in self.timesteps):
            prev_t, cur_t = t - ratio, t
            w_hist, store, w_cur, save, use = [0.0] * 4, -1, 0.0, False, False
            if counter != 1:
                ets = ets[-3:]
                store = next_slot
                # the slot being overwritten must not be one of the three kept
                while store in ets:
                    store = (store + 1) % 4
                next_slot = (store + 1) % 4
                ets = ets + [store]
            else:
                prev_t, cur_t = t, t + ratio
            if len(ets) == 1 and counter == 0:
                w_cur, save = 1.0, True
            elif len(ets) == 1 and counter == 1:
                w_cur, w_hist[ets[-1]], use = 0.5, 0.5, True
            elif len(ets) == 2:
                w_cur, w_hist[ets[-2]] = 1.5, -0.5
            elif len(ets) == 3:
                w_cur, w_hist[ets[-2]], w_hist[ets[-3]] = 23 / 12, -16 / 12, 5 / 12
            else:
                w_cur, w_hist[ets[-2]], w_hist[ets[-3]], w_hist[ets[-4]] = 55 / 24, -59 / 24, 37 / 24, -9 / 24
            cs, cm = self._coefficients(cur_t, prev_t)
            out.append((t, dict(store_slot=store, w_cur=w_cur, w_hist=w_hist, coef_sample=cs, coef_model=cm,
                                save_sample=save, use_saved=use)))
        return out
