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


class UniPCMultistepScheduler:
    """`UniPCMultistepScheduler.from_config(pipe.scheduler.config)` (run_aug/run_aug.py:218-219, `sampler="unipcmultistep"`;
    SURVEY 8f f4): the class defaults on top of the SD-1.5 config -- solver_order 2, "bh2", predict_x0, epsilon prediction,
    lower_order_final, no thresholding, "leading" spacing with steps_offset 1 (or "trailing" when built from the sdxl-turbo
    scheduler config, :223-226), final_sigmas_type "zero".  Host-side state
    only: every UniPC update (the corrector that revises the previous step with the new model output, then the
    predictor to the next timestep) is a LINEAR combination of the samples and the x0-predictions kept on the device, so
    `plan()` turns a step count into per-step coefficient rows for the fused kernel `saspa_unipc_step`:

        x0   = (x - sigma_t * eps_cfg) / alpha_t                                  (convert_model_output, on the UNcorrected x)
        x    = corr ? c_last * last + c_m0 * m0 + c_m1 * m1 + c_mt * x0 : x       (multistep_uni_c_bh_update)
        m1, m0, last = m0, x0, x
        x    = p_x * x + p_m0 * m0 + p_m1 * m1                                    (multistep_uni_p_bh_update)

    [upstream] diffusers 0.32.2 scheduling_unipc_multistep.py, recalled: parity unpinned."""

    ROW = 12            # floats per plan row (see plan())

    def __init__(self, **config):
        self.config = dict(SD15_SCHEDULER_CONFIG, solver_order=2, solver_type="bh2", predict_x0=True, lower_order_final=True,
                           final_sigmas_type="zero")
        self.config.update(config)
        c = self.config
        if c["beta_schedule"] != "scaled_linear" or c["prediction_type"] != "epsilon" or not c["predict_x0"]:
            raise NotImplementedError("only the SD-1.5 UniPC configuration (epsilon, predict_x0) is implemented")
        if c["solver_order"] not in (1, 2) or c["solver_type"] not in ("bh1", "bh2") or c["timestep_spacing"] not in ("leading", "trailing"):
            raise NotImplementedError("UniPC: solver_order <= 2, bh1 / bh2, leading / trailing spacing")
        n = c["num_train_timesteps"]
        betas = torch.linspace(c["beta_start"] ** 0.5, c["beta_end"] ** 0.5, n, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.init_noise_sigma = 1.0
        self.timesteps = None
        self.sigmas = None
        self.num_inference_steps = None

    @classmethod
    def from_config(cls, config):
        keep = {k: v for k, v in dict(config).items() if k in SD15_SCHEDULER_CONFIG or k in ("solver_order", "solver_type")}
        return cls(**keep)

    def set_timesteps(self, num_inference_steps):
        c = self.config
        n = c["num_train_timesteps"]
        self.num_inference_steps = num_inference_steps
        if c["timestep_spacing"] == "leading":
            ratio = n // (num_inference_steps + 1)
            ts = (np.arange(0, num_inference_steps + 1) * ratio).round()[::-1][:-1].copy().astype(np.int64) + c["steps_offset"]
        else:
            # "trailing": what UniPCMultistepScheduler.from_config inherits from the sdxl-turbo scheduler config
            # (run_aug/run_aug.py:223-226 with sampler = "unipcmultistep"); 2 steps -> [999, 499]
            ts = np.arange(n, 0, -n / num_inference_steps).round().copy().astype(np.int64) - 1
        ac = self.alphas_cumprod.double().numpy()
        sig = np.sqrt((1 - ac) / ac)
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        self.sigmas = np.concatenate([sig, [0.0]])          # final_sigmas_type "zero"
        self.timesteps = ts
        return ts

    @staticmethod
    def _alpha_sigma(sigma):
        alpha_t = 1.0 / np.sqrt(sigma * sigma + 1.0)
        return alpha_t, sigma * alpha_t

    def _bh(self, sig_t, sig_s0, sig_hist, order, corrector):
        """Coefficients of one uni_p / uni_c update from the sigma of the target, of the base point and of the older
        points (newest first).  Returns (c_x, c_m0, c_m1, c_mt); c_mt is the corrector's weight of the new output."""
        a_t, s_t = self._alpha_sigma(sig_t)
        a_s0, s_s0 = self._alpha_sigma(sig_s0)
        with np.errstate(divide="ignore"):
            lam_t = np.log(a_t) - np.log(s_t)
        lam_s0 = np.log(a_s0) - np.log(s_s0)
        h = lam_t - lam_s0
        rks = []
        for sg in sig_hist[:order - 1]:
            a_i, s_i = self._alpha_sigma(sg)
            rks.append((np.log(a_i) - np.log(s_i) - lam_s0) / h)
        rks.append(1.0)
        rks = np.asarray(rks, np.float64)
        hh = -h
        h_phi_1 = np.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1.0
        B_h = hh if self.config["solver_type"] == "bh1" else np.expm1(hh)
        R, bvec, fact = [], [], 1.0
        for i in range(1, order + 1):
            R.append(rks ** (i - 1))
            bvec.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1.0 / fact
        R, bvec = np.stack(R), np.asarray(bvec)
        c_x = s_t / s_s0
        c_m0 = -a_t * h_phi_1
        c_m1 = c_mt = 0.0
        if corrector:
            rhos = np.array([0.5]) if order == 1 else np.linalg.solve(R, bvec)
            c_mt = -a_t * B_h * rhos[-1]
            c_m0 += a_t * B_h * rhos[-1]
            if order == 2:
                c_m1 = -a_t * B_h * rhos[0] / rks[0]
                c_m0 += a_t * B_h * rhos[0] / rks[0]
        elif order == 2:
            rho = 0.5                                        # rhos_p = [0.5] for order 2
            c_m1 = -a_t * B_h * rho / rks[0]
            c_m0 += a_t * B_h * rho / rks[0]
        return float(c_x), float(c_m0), float(c_m1), float(c_mt)

    def plan(self, num_inference_steps=None):
        """[(t, row)] with row = [1/alpha_t, sigma_t (convert), corr flag, c_last, c_m0, c_m1, c_mt, p_x, p_m0, p_m1, 0, 0]."""
        if num_inference_steps is not None:
            self.set_timesteps(num_inference_steps)
        n_steps = len(self.timesteps)
        sg = self.sigmas
        out, lower, this_order = [], 0, 1
        for k, t in enumerate(int(v) for v in self.timesteps):
            a_k, s_k = self._alpha_sigma(sg[k])
            row = [1.0 / a_k, s_k, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
            if k > 0:                                        # corrector of the previous predictor step, at ITS order
                hist = [sg[k - 2]] if k >= 2 else []
                cx, cm0, cm1, cmt = self._bh(sg[k], sg[k - 1], hist, this_order, True)
                row[2:7] = [1.0, cx, cm0, cm1, cmt]
            order = min(self.config["solver_order"], n_steps - k) if self.config["lower_order_final"] else self.config["solver_order"]
            this_order = min(order, lower + 1)
            hist = [sg[k - 1]] if k >= 1 else []
            px, pm0, pm1, _ = self._bh(sg[k + 1], sg[k], hist, this_order, False)
            row[7:10] = [px, pm0, pm1]
            if lower < self.config["solver_order"]:
                lower += 1
            out.append((t, row))
        return out
