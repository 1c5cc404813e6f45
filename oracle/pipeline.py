"""TEST INFRASTRUCTURE ONLY -- torch-CPU fp32 restatement of
``StableDiffusionControlNetPipeline.__call__`` exactly as the reference invokes
it (run_aug/run_aug.py:235-241, :268-269, :278): CFG with a negative prompt,
Canny control image, DDIM (scheduler built at run_aug/run_aug.py:221 from the
SD-1.5 config), eta=0, conditioning scale 0.75, output_type "pil".

PARITY UNPINNED (see oracle/__init__.py).
"""
import numpy as np
import torch

from . import sd_models as M


class DDIM:
    """DDIMScheduler.from_config(SD-1.5 scheduler config): scaled_linear betas
    0.00085..0.012 over 1000 steps, steps_offset=1, set_alpha_to_one=False,
    clip_sample=False, epsilon prediction, "leading" spacing."""

    def __init__(self, num_train=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1, spacing="leading"):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.num_train = num_train
        self.steps_offset = steps_offset
        self.spacing = spacing
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n):
        self.n = n
        if self.spacing == "trailing":      # SDXL-Turbo's scheduler config (timestep_spacing="trailing")
            ts = np.round(np.arange(self.num_train, 0, -self.num_train / n)).astype(np.int64) - 1
        else:
            ratio = self.num_train // n
            ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.timesteps = ts
        return ts

    def coefficients(self, t):
        prev_t = int(t) - self.num_train // self.n
        a_t = self.alphas_cumprod[int(t)]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return a_t, a_prev

    def step(self, eps, t, x):
        a_t, a_prev = self.coefficients(t)
        beta_t = 1 - a_t
        x0 = (x - beta_t ** 0.5 * eps) / a_t ** 0.5
        direction = (1 - a_prev) ** 0.5 * eps          # eta = 0 -> std_dev_t = 0
        return a_prev ** 0.5 * x0 + direction


class PNDM:
    """PNDMScheduler as shipped in the SD-1.5-derived BLIP-Diffusion repo (skip_prk_steps=True, scaled_linear
    0.00085..0.012, steps_offset=1, set_alpha_to_one=False, "leading"): ``step_plms`` restated with the
    reference's ``ets`` list ([upstream] diffusers 0.32.2 scheduling_pndm.py, recalled)."""

    def __init__(self, num_train=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.num_train, self.steps_offset, self.init_noise_sigma = num_train, steps_offset, 1.0

    def set_timesteps(self, n):
        self.n = n
        ratio = self.num_train // n
        base = (np.arange(0, n) * ratio).round().astype(np.int64) + self.steps_offset
        self.timesteps = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy()
        self.ets, self.counter, self.cur_sample = [], 0, None
        return self.timesteps

    def _get_prev_sample(self, sample, timestep, prev_timestep, model_output):
        a_t = self.alphas_cumprod[timestep]
        a_p = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b_t, b_p = 1 - a_t, 1 - a_p
        sample_coeff = (a_p / a_t) ** 0.5
        denom = a_t * b_p ** 0.5 + (a_t * b_t * a_p) ** 0.5
        return sample_coeff * sample - (a_p - a_t) * model_output / denom

    def step(self, model_output, timestep, sample):
        timestep = int(timestep)
        prev_timestep = timestep - self.num_train // self.n
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev_timestep = timestep
            timestep = timestep + self.num_train // self.n
        if len(self.ets) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(self.ets) == 1 and self.counter == 1:
            model_output = (model_output + self.ets[-1]) / 2
            sample = self.cur_sample
            self.cur_sample = None
        elif len(self.ets) == 2:
            model_output = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif len(self.ets) == 3:
            model_output = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            model_output = (1 / 24) * (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4])
        prev_sample = self._get_prev_sample(sample, timestep, prev_timestep, model_output)
        self.counter += 1
        return prev_sample


class UniPC:
    """UniPCMultistepScheduler with the class defaults on the SD-1.5 config (solver_order 2, bh2, predict_x0, epsilon,
    lower_order_final, leading spacing + steps_offset 1, final sigma zero): a STATEFUL restatement of set_timesteps /
    step / convert_model_output / multistep_uni_p_bh_update / multistep_uni_c_bh_update ([upstream] diffusers 0.32.2
    scheduling_unipc_multistep.py, recalled).  The product derives closed-form per-step coefficients instead
    (saspa_aug_amd.scheduler.UniPCMultistepScheduler.plan); tests compare the two on arbitrary model outputs."""

    def __init__(self, num_train=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1, solver_order=2, solver_type="bh2",
                 spacing="leading"):
        self.spacing = spacing                               # "trailing": built from the sdxl-turbo scheduler config
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.num_train, self.steps_offset, self.order, self.solver_type = num_train, steps_offset, solver_order, solver_type
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n):
        if self.spacing == "trailing":
            ts = np.arange(self.num_train, 0, -self.num_train / n).round().copy().astype(np.int64) - 1
        else:
            ratio = self.num_train // (n + 1)
            ts = (np.arange(0, n + 1) * ratio).round()[::-1][:-1].copy().astype(np.int64) + self.steps_offset
        ac = self.alphas_cumprod.double().numpy()
        sig = np.sqrt((1 - ac) / ac)
        self.sigmas = torch.from_numpy(np.concatenate([np.interp(ts, np.arange(0, len(sig)), sig), [0.0]]))
        self.timesteps = ts
        self.model_outputs = [None] * self.order
        self.lower_order_nums, self.step_index, self.last_sample, self.this_order = 0, 0, None, 1
        return ts

    @staticmethod
    def _as(sigma):
        alpha_t = 1 / (sigma ** 2 + 1) ** 0.5
        return alpha_t, sigma * alpha_t

    def _update(self, x, m0, older, sig_t, sig_s0, sig_older, order, model_t=None):
        a_t, s_t = self._as(sig_t)
        a_s0, s_s0 = self._as(sig_s0)
        lam_t, lam_s0 = torch.log(a_t) - torch.log(s_t), torch.log(a_s0) - torch.log(s_s0)
        h = lam_t - lam_s0
        rks, D1s = [], []
        for mi, sg in zip(older[:order - 1], sig_older[:order - 1]):
            a_i, s_i = self._as(sg)
            rk = (torch.log(a_i) - torch.log(s_i) - lam_s0) / h
            rks.append(rk)
            D1s.append((mi - m0) / rk)
        rks.append(torch.tensor(1.0, dtype=torch.float64))
        rks = torch.stack([torch.as_tensor(r, dtype=torch.float64) for r in rks])
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = hh if self.solver_type == "bh1" else torch.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        R, b = torch.stack(R), torch.stack([torch.as_tensor(v, dtype=torch.float64) for v in b])
        x_t_ = s_t / s_s0 * x - a_t * h_phi_1 * m0
        if model_t is None:                                  # predictor
            res = 0
            if D1s:
                rhos = torch.tensor([0.5], dtype=torch.float64) if order == 2 else torch.linalg.solve(R[:-1, :-1], b[:-1])
                res = sum(r * d for r, d in zip(rhos, D1s))
            return x_t_ - a_t * B_h * res
        rhos = torch.tensor([0.5], dtype=torch.float64) if order == 1 else torch.linalg.solve(R, b)
        res = sum(r * d for r, d in zip(rhos[:-1], D1s)) if D1s else 0
        return x_t_ - a_t * B_h * (res + rhos[-1] * (model_t - m0))

    def step(self, eps, t, sample):
        k = self.step_index
        sg = self.sigmas
        a_t, s_t = self._as(sg[k])
        x0 = (sample.double() - s_t * eps.double()) / a_t
        if k > 0 and self.last_sample is not None:
            older = [self.model_outputs[-2]] if k >= 2 else []
            sample = self._update(self.last_sample, self.model_outputs[-1], older, sg[k], sg[k - 1], [sg[k - 2]] if k >= 2 else [],
                                  self.this_order, model_t=x0)
        self.model_outputs = self.model_outputs[1:] + [x0]
        order = min(self.order, len(self.timesteps) - k)
        self.this_order = min(order, self.lower_order_nums + 1)
        self.last_sample = sample.double()
        older = [self.model_outputs[-2]] if k >= 1 else []
        prev = self._update(self.last_sample, x0, older, sg[k + 1], sg[k], [sg[k - 1]] if k >= 1 else [], self.this_order)
        if self.lower_order_nums < self.order:
            self.lower_order_nums += 1
        self.step_index += 1
        return prev.float()


def prepare_control(control_u8):
    """VaeImageProcessor(do_normalize=False).preprocess: u8 HWC RGB -> [1,3,H,W] in [0,1]."""
    x = torch.from_numpy(np.ascontiguousarray(control_u8)).float() / 255.0
    return x.permute(2, 0, 1)[None]


def postprocess(img):
    """VaeImageProcessor.postprocess('pil') up to the uint8 array."""
    x = (img / 2 + 0.5).clamp(0, 1)
    x = x.permute(0, 2, 3, 1).float().numpy()
    return (x * 255).round().astype("uint8")


@torch.no_grad()
def sd_controlnet_pipeline(weights, cfgs, ids_pos, ids_neg, control_u8, latents, steps,
                           guidance_scale=7.5, conditioning_scale=0.75, return_latents=False, trace=None, sampler="ddim"):
    """weights: dict(unet=, controlnet=, vae=, text=) of diffusers-named state dicts.
    ids_*: int64 [1,77].  control_u8: u8 [H,W,3].  latents: fp32 [1,4,H/8,W/8] noise.
    Returns u8 [1,H,W,3] (and the final latents / decoded float image if asked).  `trace` (a list) receives one
    dict(t, x, eps2, ctx) per step: the latents going INTO the evaluation and the two CFG halves coming out of it
    (teacher-forced per-evaluation checks of the bf16 path)."""
    ctx = M.clip_text_forward(weights["text"], cfgs["text"], torch.cat([ids_neg, ids_pos], 0))
    cond = prepare_control(control_u8)
    cond2 = torch.cat([cond, cond], 0)
    sch = DDIM() if sampler == "ddim" else UniPC()
    x = latents.clone().float() * sch.init_noise_sigma
    for t in sch.set_timesteps(steps):
        x2 = torch.cat([x, x], 0)
        down, mid = M.controlnet_forward(weights["controlnet"], cfgs["controlnet"], x2, int(t), ctx, cond2,
                                         conditioning_scale)
        eps2 = M.unet_forward(weights["unet"], cfgs["unet"], x2, int(t), ctx, down, mid)
        if trace is not None:
            trace.append(dict(t=int(t), x=x.clone(), eps2=eps2.clone(), ctx=ctx))
        eps_u, eps_c = eps2.chunk(2)
        eps = eps_u + guidance_scale * (eps_c - eps_u)
        x = sch.step(eps, t, x)
    img = M.vae_decode(weights["vae"], cfgs["vae"], x / cfgs["vae"]["scaling_factor"])
    out = postprocess(img)
    if "safety" in weights and "safety" in cfgs:
        out = run_safety_checker(weights["safety"], cfgs["safety"], out)[0]
    if return_latents:
        return out, x, img
    return out


@torch.no_grad()
def sd_controlnet_img2img_pipeline(weights, cfgs, ids_pos, ids_neg, source_u8, control_u8, sample_noise, noise, steps, strength,
                                   guidance_scale=7.5, conditioning_scale=0.75, return_latents=False):
    """StableDiffusionControlNetImg2ImgPipeline.__call__ as the reference invokes it with SDEDIT = 1
    (run_aug/run_aug.py:203-206, :252-260, :274-276; [upstream] diffusers 0.32.2): the source image (VaeImageProcessor:
    [-1, 1]) is encoded, latents = (mean + exp(0.5 * clamp(logvar, -30, 20)) * sample_noise) * scaling_factor, noised to
    the first kept timestep with `noise` (scheduler.add_noise), then the LAST int(steps * strength) of the `steps` DDIM
    steps run with CFG + ControlNet.  sample_noise / noise: [1,4,H/8,W/8], the two consecutive draws from the generator."""
    ctx = M.clip_text_forward(weights["text"], cfgs["text"], torch.cat([ids_neg, ids_pos], 0))
    cond = prepare_control(control_u8)
    cond2 = torch.cat([cond, cond], 0)
    src = torch.from_numpy(np.ascontiguousarray(source_u8)).float().permute(2, 0, 1)[None] / 127.5 - 1.0
    mean, logvar = M.vae_encode(weights["vae"], cfgs["vae"], src)
    std = torch.exp(0.5 * logvar.clamp(-30.0, 20.0))
    x0 = (mean + std * sample_noise.float()) * cfgs["vae"]["scaling_factor"]
    sch = DDIM()
    ts_all = sch.set_timesteps(steps)
    init = min(int(steps * strength), steps)
    ts = ts_all[max(steps - init, 0):]
    a_t = sch.alphas_cumprod[int(ts[0])]
    x = a_t ** 0.5 * x0 + (1 - a_t) ** 0.5 * noise.float()
    for t in ts:
        x2 = torch.cat([x, x], 0)
        down, mid = M.controlnet_forward(weights["controlnet"], cfgs["controlnet"], x2, int(t), ctx, cond2, conditioning_scale)
        eps2 = M.unet_forward(weights["unet"], cfgs["unet"], x2, int(t), ctx, down, mid)
        eps_u, eps_c = eps2.chunk(2)
        x = sch.step(eps_u + guidance_scale * (eps_c - eps_u), t, x)
    img = M.vae_decode(weights["vae"], cfgs["vae"], x / cfgs["vae"]["scaling_factor"])
    out = postprocess(img)
    if return_latents:
        return out, x, img
    return out


@torch.no_grad()
def sd_img2img_pipeline(weights, cfgs, ids_pos, ids_neg, source_u8, sample_noise, noise, steps, strength, guidance_scale=7.5,
                        return_latents=False):
    """StableDiffusionImg2ImgPipeline.__call__ -- the reference's CONTROLNET = None, SDEDIT = 1 branch
    (run_aug/run_aug.py:163-165, :235-241, :274-276; Real-Guidance defaults run_aug/run_aug_real_guidance.py:520-523;
    [upstream] diffusers 0.32.2): the img2img procedure of `sd_controlnet_img2img_pipeline` with every evaluation the UNet
    alone.  PARITY UNPINNED (diffusers absent), like the ControlNet form it restates next to."""
    ctx = M.clip_text_forward(weights["text"], cfgs["text"], torch.cat([ids_neg, ids_pos], 0))
    src = torch.from_numpy(np.ascontiguousarray(source_u8)).float().permute(2, 0, 1)[None] / 127.5 - 1.0
    mean, logvar = M.vae_encode(weights["vae"], cfgs["vae"], src)
    std = torch.exp(0.5 * logvar.clamp(-30.0, 20.0))
    x0 = (mean + std * sample_noise.float()) * cfgs["vae"]["scaling_factor"]
    sch = DDIM()
    ts_all = sch.set_timesteps(steps)
    init = min(int(steps * strength), steps)
    ts = ts_all[max(steps - init, 0):]
    a_t = sch.alphas_cumprod[int(ts[0])]
    x = a_t ** 0.5 * x0 + (1 - a_t) ** 0.5 * noise.float()
    for t in ts:
        eps2 = M.unet_forward(weights["unet"], cfgs["unet"], torch.cat([x, x], 0), int(t), ctx)
        eps_u, eps_c = eps2.chunk(2)
        x = sch.step(eps_u + guidance_scale * (eps_c - eps_u), t, x)
    img = M.vae_decode(weights["vae"], cfgs["vae"], x / cfgs["vae"]["scaling_factor"])
    out = postprocess(img)
    if return_latents:
        return out, x, img
    return out


def run_safety_checker(sd, cfg, images_u8):
    """StableDiffusionControlNetPipeline.run_safety_checker + the checker's black-out: u8 [B,H,W,3] ->
    (u8 images with flagged ones zeroed, flags)."""
    from . import image_ops as IO
    px = torch.stack([IO.clip_image_preprocess(im, cfg["image_size"]) for im in images_u8])
    flags, _, _ = IO.safety_checker_forward(sd, cfg, px)
    out = images_u8.copy()
    for i, f in enumerate(flags):
        if f:
            out[i] = 0
    return out, flags


def build_blip_prompt(prompt, tgt_subject, prompt_strength=1.0, prompt_reps=20):
    """BlipDiffusionControlNetPipeline._build_prompt: "a {subject} {prompt}" repeated (the prompt amplifier)."""
    p = f"a {tgt_subject} {prompt.strip()}"
    return ", ".join([p] * int(prompt_strength * prompt_reps))


@torch.no_grad()
def blip_controlnet_pipeline(weights, cfgs, ids_prompt, ids_neg, query_embeds, control_u8, latents, steps,
                             guidance_scale=7.5, ctx_begin_pos=2, return_latents=False):
    """BlipDiffusionControlNetPipeline.__call__ as the reference invokes it (run_aug/run_aug.py:243-250, 262-265):
    ids_prompt int64 [1, 77-nq] (the amplified prompt, tokenised to max_length - num_query_tokens), ids_neg [1,77],
    query_embeds fp32 [1, nq, width] (Q-Former subject tokens), ControlNet conditioning scale 1.0 (none is passed),
    PNDM/PLMS scheduler kept.  Returns u8 [1,H,W,3]."""
    pos = M.clip_text_forward(weights["text"], cfgs["text"], ids_prompt, query_embeds, ctx_begin_pos)
    neg = M.clip_text_forward(weights["text"], cfgs["text"], ids_neg)
    ctx = torch.cat([neg, pos], 0)
    cond = prepare_control(control_u8)
    cond2 = torch.cat([cond, cond], 0)
    sch = PNDM()
    x = latents.clone().float() * sch.init_noise_sigma
    for t in sch.set_timesteps(steps):
        x2 = torch.cat([x, x], 0)
        down, mid = M.controlnet_forward(weights["controlnet"], cfgs["controlnet"], x2, int(t), ctx, cond2, 1.0)
        eps2 = M.unet_forward(weights["unet"], cfgs["unet"], x2, int(t), ctx, down, mid)
        eps_u, eps_c = eps2.chunk(2)
        eps = eps_u + guidance_scale * (eps_c - eps_u)
        x = sch.step(eps, t, x)
    img = M.vae_decode(weights["vae"], cfgs["vae"], x / cfgs["vae"]["scaling_factor"])
    out = postprocess(img)
    if return_latents:
        return out, x, img
    return out


@torch.no_grad()
def sdxl_controlnet_pipeline(weights, cfgs, ids1, ids2, control_u8, latents, steps, conditioning_scale=0.75,
                             return_latents=False, guidance_scale=0.0, neg_ids1=None, neg_ids2=None, sampler="ddim"):
    """StableDiffusionXLControlNetPipeline.__call__ as the reference invokes it for sd_xl-turbo
    (run_aug/run_aug.py:189-201, :223-228, :564-571): guidance_scale 0 -> NO classifier-free guidance (one
    conditional evaluation per step, no negative prompt), 2 steps, DDIMScheduler.from_config(<SDXL-Turbo scheduler
    config>) -> "trailing" timesteps, set_alpha_to_one=False, clip_sample=False; conditioning scale 0.75 (:269);
    VAE = sdxl-vae-fp16-fix upcast to fp32, scaling factor 0.13025.
    ids1 / ids2: int64 [1,77] from tokenizer (EOS-padded) / tokenizer_2 (0-padded).  Prompt embedding = concat of the
    two towers' hidden_states[-2]; pooled = text_encoder_2's projected EOS state; add_time_ids = (H, W, 0, 0, H, W).
    guidance_scale > 1 (BASELINE configs[4] family): CFG, uncond half first; without a negative prompt the negative
    embeddings are zeros (force_zeros_for_empty_prompt of the sdxl-turbo repo)."""
    h1, _ = M.clip_text_forward(weights["text"], cfgs["text"], ids1, penultimate=True)
    h2, pooled = M.clip_text_forward(weights["text2"], cfgs["text2"], ids2, penultimate=True)
    ctx = torch.cat([h1, h2], dim=-1)
    hh, ww = control_u8.shape[:2]
    cfg = guidance_scale > 1.0
    cond = prepare_control(control_u8)
    if cfg:
        if neg_ids1 is None:
            nctx, npooled = torch.zeros_like(ctx), torch.zeros_like(pooled)
        else:
            n1, _ = M.clip_text_forward(weights["text"], cfgs["text"], neg_ids1, penultimate=True)
            n2, npooled = M.clip_text_forward(weights["text2"], cfgs["text2"], neg_ids2, penultimate=True)
            nctx = torch.cat([n1, n2], dim=-1)
        ctx = torch.cat([nctx, ctx], 0)
        pooled = torch.cat([npooled, pooled], 0)
        cond = torch.cat([cond, cond], 0)
    nb = ctx.shape[0]
    added = dict(text_embeds=pooled, time_ids=torch.tensor([[hh, ww, 0, 0, hh, ww]] * nb, dtype=torch.float32))
    # sampler "unipc": UniPCMultistepScheduler.from_config(<the same scheduler config>) (run_aug/run_aug.py:223-226)
    sch = UniPC(spacing="trailing") if sampler == "unipc" else DDIM(spacing="trailing")
    x = latents.clone().float() * sch.init_noise_sigma
    for t in sch.set_timesteps(steps):
        xin = torch.cat([x, x], 0) if cfg else x
        down, mid = M.controlnet_forward(weights["controlnet"], cfgs["controlnet"], xin, int(t), ctx, cond,
                                         conditioning_scale, added)
        eps = M.unet_forward(weights["unet"], cfgs["unet"], xin, int(t), ctx, down, mid, added)
        if cfg:
            eps_u, eps_c = eps.chunk(2)
            eps = eps_u + guidance_scale * (eps_c - eps_u)
        x = sch.step(eps, t, x)
    img = M.vae_decode(weights["vae"], cfgs["vae"], x / cfgs["vae"]["scaling_factor"])
    out = postprocess(img)
    if return_latents:
        return out, x, img
    return out
