"""Drop-in for the diffusers pipeline object the reference builds in `init_pipeline`
(run_aug/run_aug.py:128-230) and calls at run_aug/run_aug.py:278:

    pipe = StableDiffusionControlNetPipeline.from_pretrained(...).to(DEVICE, torch.float16)
    pipe.scheduler = DDIMScheduler.from_config(pipe.scheduler.config)
    image = pipe(prompt=..., image=<canny PIL>, num_inference_steps=..., generator=...,
                 guidance_scale=7.5, negative_prompt=..., controlnet_conditioning_scale=0.75).images[0]

Same names, argument meaning and error behaviour (Python exceptions; device problems are
RuntimeError, which is what the reference's loop catches, run_aug/run_aug.py:493).  The whole
sampling loop runs in the gfx950 kernels; `generate_batch` is the batched form (B images per
launch sequence, CFG doubles it) that run_aug and bench.py drive.

Precision: `.to(device, torch.float16)` / `torch.bfloat16` selects the bf16 MFMA path (the
MI355X counterpart of the reference's fp16 CUDA path; bf16 also avoids the SD-VAE fp16
overflow); `.to(device, torch.float32)` selects the exact-fp32 MFMA path used for the
end-to-end parity gate against the CPU oracle."""
import os

import numpy as np
import torch
from PIL import Image

from . import models, ops
from . import weights as W
from .config import BLIP_DIFFUSION, BLIP_IMAGE_MEAN, BLIP_IMAGE_STD, SD15, SDXL_TURBO
from .scheduler import SDXL_TURBO_SCHEDULER_CONFIG, DDIMScheduler, PNDMScheduler, UniPCMultistepScheduler
from .tokenizer import make_bert_tokenizer, make_tokenizer


class PipelineOutput:
    def __init__(self, images, nsfw_content_detected=None):
        self.images = images
        self.nsfw_content_detected = nsfw_content_detected


def graphs_enabled():
    """hipGraph replay of the DDIM sampling step (default on; SASPA_GRAPH=0 launches every kernel from Python).  Off while
    a launch recorder is installed (bench.py times individual launches with HIP events)."""
    return os.environ.get("SASPA_GRAPH", "1") != "0" and ops._RECORDER is None


def fork_enabled():
    """Two-branch sampling step (UNet encoder || ControlNet encoder inside the captured graph).  SASPA_FORK=0 captures
    the single-stream order."""
    return os.environ.get("SASPA_FORK", "1") != "0"


def replay_queue_depth():
    """Replays of the step graph the host keeps outstanding before it sleeps until the oldest has finished (see _StepGraph.run;
    0 = unthrottled)."""
    try:
        return int(os.environ.get("SASPA_REPLAY_DEPTH", "3"))
    except ValueError:
        return 3


side_stream = ops.side_stream          # one capture side stream per process and device (ops.side_stream)


class _StepGraph:
    """ONE captured hipGraph of a sampling step -- UNet encoder, ControlNet, UNet decoder, (CFG +) DDIM or PLMS update,
    ~1 500 kernel nodes -- replayed once per network evaluation.  Launching those kernels from Python costs 1.28 s per batch-8 / 50-step
    generation against 1.38 s of GPU time (tools/host_launch_time.py): the host was 7 % away from being the bottleneck.
    Everything a step reads lives in static buffers owned by this object; what changes from step to step is read on the
    device through a step counter (time-embedding rows: ops.gather_row; DDIM coefficients: ops.ddim_step_dev)."""

    def __init__(self, pipe, x_shape, cemb_shape, steps, cfg, guidance, cscale):
        dev, dt = pipe.device, pipe.dtype
        self.pipe, self.steps, self.cfg, self.guidance, self.cscale = pipe, steps, cfg, guidance, cscale
        self.x = torch.zeros(x_shape, device=dev, dtype=dt)
        self.eps = torch.zeros(x_shape, device=dev, dtype=dt)
        self.cemb = torch.zeros(cemb_shape, device=dev, dtype=dt)
        self.idx = torch.zeros((1,), device=dev, dtype=torch.int32)
        self.plms = isinstance(pipe.scheduler, PNDMScheduler)
        self.unipc = isinstance(pipe.scheduler, UniPCMultistepScheduler)
        if self.unipc:      # per-step coefficient rows; last sample + two x0-predictions in one static buffer
            self.evals = steps
            self.coefs = torch.zeros((steps, UniPCMultistepScheduler.ROW), device=dev, dtype=torch.float32)
            self.state = torch.zeros((3, x_shape[0] // 2 if cfg else x_shape[0]) + tuple(x_shape[1:]), device=dev, dtype=dt)
        elif self.plms:       # N + 1 evaluations; per-evaluation parameters, the 4-slot history ring and the saved sample
            self.evals = steps + 1
            self.coefs = torch.zeros((self.evals, 10), device=dev, dtype=torch.float32)
            self.hist = torch.zeros((4, x_shape[0] // 2) + tuple(x_shape[1:]), device=dev, dtype=dt)
            self.saved = torch.zeros((x_shape[0] // 2,) + tuple(x_shape[1:]), device=dev, dtype=dt)
        else:
            self.evals = steps
            self.coefs = torch.zeros((steps, 4), device=dev, dtype=torch.float32)
        self.nets = (pipe.unet,) if pipe.controlnet is None else (pipe.unet, pipe.controlnet)   # ControlNet-free: img2img baseline
        self.tables, self.curs, self.ctx_kv = [], [], []
        self.graph = None
        self.side = None                  # second capture stream of the forked step (fork_enabled)
        self._replay_events = []          # events of the replays still in the queue (run())

    def _bind(self):
        """Point the networks at this graph's static state."""
        for net, cur, kv in zip(self.nets, self.curs, self.ctx_kv):
            net.bind_step_state(cur)
            net.ctx_kv = kv

    def load(self, x, cemb, ctx, ts, added=None):
        """Refresh the static buffers for one generation (device-side copies; shapes are fixed by the cache key)."""
        sch = self.pipe.scheduler
        first = not self.tables
        refresh_t = first or added is not None          # without SDXL's added conditioning the time tables depend on ts only
        for i, net in enumerate(self.nets):
            net.prepare_context(ctx)
            if refresh_t:
                net.prepare_timesteps(ts, added)
            if first:
                self.tables.append(net.temb_all.clone())
                self.curs.append(torch.zeros_like(net.temb_all[0]))
                # (K, V^T, key count[, K / V^T fragments of the fused cross-attention launch]): tensors are cloned into static buffers
                self.ctx_kv.append({t: tuple(e.clone() if torch.is_tensor(e) else e for e in kv) for t, kv in net.ctx_kv.items()})
            else:
                if refresh_t:
                    self.tables[i].copy_(net.temb_all)
                for t, kv in net.ctx_kv.items():
                    for dst, src in zip(self.ctx_kv[i][t], kv):
                        if torch.is_tensor(src):
                            dst.copy_(src)
        self.x.copy_(x)
        self.cemb.copy_(cemb)
        if self.unipc:
            if first:
                self.coefs.copy_(torch.tensor(self.plan, dtype=torch.float32))
            self.state.zero_()
        elif self.plms:
            if first:
                rows = [[d["store_slot"], d["w_cur"], *d["w_hist"], d["coef_sample"], d["coef_model"], float(d["save_sample"]),
                         float(d["use_saved"])] for d in self.plan]
                self.coefs.copy_(torch.tensor(rows, dtype=torch.float32))
            self.hist.zero_()
        elif first:
            self.coefs.copy_(torch.tensor([sch.step_coefficients(t) for t in ts], dtype=torch.float32))
        self.idx.zero_()
        self._bind()

    def _step(self):
        pipe, x = self.pipe, self.x
        for net, tab, cur in zip(self.nets, self.tables, self.curs):
            ops.gather_row(tab, self.idx, cur)
        if pipe.controlnet is None:
            mid2, skips2 = pipe.unet.encode(x, None)
        elif fork_enabled():
            # the UNet encoder and the ControlNet encoder are independent until the zero convs add the two: two
            # branches of the captured graph.  Same kernels on the same inputs -> bit-identical to the single-stream order; what
            # changes is that one branch's launch gaps / tile tails / small deep-level kernels are filled by the other's work.
            main = torch.cuda.current_stream()
            if self.side is None:
                self.side = side_stream(x.device)
            self.side.wait_stream(main)
            with ops.twin_branch(ops._RECORDER is None):   # both encoders walk the same shapes side by side: half the K slices each
                with torch.cuda.stream(self.side):
                    cmid, cfeats = pipe.controlnet.encode(x, None, conv_in_residual=self.cemb)
                mid, skips = pipe.unet.encode(x, None)
            main.wait_stream(self.side)
            skips2, mid2 = pipe.controlnet.zero_convs(cmid, cfeats, self.cscale, skips, mid)
        else:
            mid, skips = pipe.unet.encode(x, None)
            skips2, mid2 = pipe.controlnet.forward(x, None, self.cemb, self.cscale, skips, mid)
        pipe.unet.decode(mid2, skips2, None, out=self.eps)
        nimg = x.shape[0] // 2 if self.cfg else x.shape[0]
        nc, hw = pipe.cfgs["unet"]["out_channels"], x.shape[1] * x.shape[2]
        if self.unipc and self.cfg:
            ops.cfg_unipc_step(self.eps, x, self.state, nimg, hw, nc, self.guidance, table=self.coefs, index=self.idx)
        elif self.unipc:
            ops.unipc_step(self.eps, x, self.state, nimg, hw, nc, table=self.coefs, index=self.idx)
        elif self.plms:
            ops.cfg_plms_step_dev(self.eps, x, self.hist, self.saved, nimg, hw, nc, self.guidance, self.coefs, self.idx)
        else:
            ops.ddim_step_dev(self.eps, x, nimg, hw, nc, self.guidance, self.coefs, self.idx, cfg=self.cfg)
        ops.index_add(self.idx, 1)

    def run_on(self, x, cemb, ctx, ts, added=None, plan=None):
        self.plan = plan
        self.load(x, cemb, ctx, ts, added)
        return self.run()

    def run(self):
        """All steps: the first generation on this object runs step 0 eagerly (warms allocator / lazy state), restores
        the inputs, captures the step and replays; later generations only replay."""
        if self.graph is None:
            x0 = self.x.clone()
            self._step()                                   # eager warm-up step (also validates every launch argument)
            torch.cuda.synchronize()
            self.x.copy_(x0)
            self.idx.zero_()
            if self.plms:
                self.hist.zero_()
            if self.unipc:
                self.state.zero_()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._step()
            self.graph = g
        depth = replay_queue_depth()
        if depth <= 0:
            for _ in range(self.evals):
                self.graph.replay()
            return self.x
        # Host throttle (round 6): the HIP queue holds about five replays of the ~1 500-node step; once it is full, hipGraphLaunch
        # SPINS on the calling thread until a slot frees -- one host core per rank at 100 % for the whole run
        # (profiles/r6_config3_full_shard.json: 0.21 CPU-s per image in the launch thread alone; a replay itself costs the host
        # 1.4 ms, tools/graph_host_cost.py).  Keeping at most `depth` replays outstanding and SLEEPING until the oldest has finished
        # (ops.sleep_wait: event query + 1 ms naps -- hipEventSynchronize busy-waits here even with hipEventBlockingSync) leaves
        # the GPU the same backlog to chew on (3 replays = 70-100 ms) and gives the core back.  SASPA_REPLAY_DEPTH=0 = unthrottled.
        evs = self._replay_events
        for _ in range(self.evals):
            if len(evs) >= depth:
                ops.sleep_wait(evs.pop(0))
            self.graph.replay()
            e = torch.cuda.Event()
            e.record()
            evs.append(e)
        return self.x


class StableDiffusionControlNetPipeline:
    HAS_CONTROLNET = True

    def __init__(self, state_dicts, cfgs=SD15, tokenizer=None, scheduler=None):
        self._state_dicts = state_dicts
        self.cfgs = cfgs
        self.scheduler = scheduler or DDIMScheduler()
        self.tokenizer = tokenizer or make_tokenizer(vocab=cfgs["text"]["vocab"])
        self.device = None
        self.dtype = None
        self.noise_dtype = None
        self.unet = self.controlnet = self.vae = self.text_encoder = None
        self._graphs = {}                 # (shapes, steps, guidance, scale) -> _StepGraph, small LRU
        self.safety_checker = None        # built by .to() when the family ships one; assign None to disable (diffusers idiom)
        self.last_nsfw = None
        self._neg_cache = {}

    # ---- construction -------------------------------------------------------------------
    @classmethod
    def from_synthetic(cls, cfgs=SD15, seed=0):
        """Architecture-exact random weights (no checkpoint / network available)."""
        return cls(W.synth_family(cfgs, seed), cfgs)

    @classmethod
    def from_pretrained(cls, base_dir, controlnet_dir, cfgs=SD15):
        """Local diffusers-format checkpoints (the layout `from_pretrained` downloads):
        {base}/unet, {base}/vae, {base}/text_encoder, {base}/tokenizer and the ControlNet dir."""
        def f(d, *names):
            for n in names:
                p = os.path.join(d, n)
                if os.path.exists(p):
                    return W.load_safetensors(p)
            raise FileNotFoundError(f"no safetensors weights in {d}")
        sds = dict(
            unet=f(os.path.join(base_dir, "unet"), "diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors"),
            vae=f(os.path.join(base_dir, "vae"), "diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors"),
            text=f(os.path.join(base_dir, "text_encoder"), "model.safetensors", "model.fp16.safetensors"),
            controlnet=f(controlnet_dir, "diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors"),
        )
        sc = os.path.join(base_dir, "safety_checker")
        if "safety" in cfgs and os.path.isdir(sc):
            sds["safety"] = f(sc, "model.safetensors", "model.fp16.safetensors")
        return cls(sds, cfgs, tokenizer=make_tokenizer(os.path.join(base_dir, "tokenizer"), cfgs["text"]["vocab"]))

    def to(self, device, dtype=None):
        device = torch.device(device)
        if self._state_dicts is None:
            raise RuntimeError("this pipeline was already placed on a device; build a new one to change device / dtype")
        if device.type != "cuda":
            raise RuntimeError("saspa_aug_amd pipelines run on an MI355X only (no CPU path); got device %s" % device)
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible")
        if dtype in (None, torch.float16, torch.bfloat16):
            cdt, ndt = torch.bfloat16, torch.float16   # noise is drawn in the reference's pipeline dtype
        elif dtype == torch.float32:
            cdt, ndt = torch.float32, torch.float32
        else:
            raise TypeError(f"unsupported pipeline dtype {dtype}")
        self.device, self.dtype, self.noise_dtype = device, cdt, ndt
        # the kernels launch on the CURRENT HIP device / torch's current stream of it: make the pipeline's device current
        # (the reference's idiom is editing DEVICE = "cuda:1"; without this tensors would live on GPU 1, launches on GPU 0)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        torch.cuda.set_device(device)
        self.device = device
        sd, cf = self._state_dicts, self.cfgs
        fp8 = getattr(self, "_fp8", False) or os.environ.get("SASPA_FP8", "0") == "1"
        self.unet = models.UNet(sd["unet"], cf["unet"], device, cdt, fp8=fp8)
        self.controlnet = models.ControlNet(sd["controlnet"], cf["controlnet"], device, cdt, fp8=fp8) if self.HAS_CONTROLNET else None
        self.vae = models.VAEDecoder(sd["vae"], cf["vae"], device, cdt)
        self.text_encoder = models.CLIPText(sd["text"], cf["text"], device, cdt)
        self._build_extra(sd, cf, device, cdt)
        self._neg_cache = {}
        self._state_dicts = None      # the packed device copies are the weights now; free ~5.6 GB of host fp32
        return self

    def _build_extra(self, sd, cf, device, cdt):
        # StableDiffusionSafetyChecker of the SD-1.5 repo: the reference never passes safety_checker=None
        if "safety" in sd and "safety" in cf:
            self.safety_checker = models.SafetyChecker(sd["safety"], cf["safety"], device, cdt)

    def enable_fp8(self, on=True):
        """Call BEFORE `.to()`: the LayerNorm-fed projections of the UNet / ControlNet transformer blocks (cross-attention
        query, GEGLU feed-forward) run as e4m3 W8A8 GEMMs (BASELINE.json configs[4] "fp8 MFMA"; SASPA_FP8=1 does the same).
        Off by default: the headline metric is quoted in bf16."""
        if self.unet is not None:
            raise RuntimeError("enable_fp8() must be called before .to(): the weights are quantised at pack time")
        self._fp8 = bool(on)
        return self

    def upcast_vae(self, gemm=None):  # SDXL-only hook the reference calls at run_aug/run_aug.py:224
        return self

    def run_safety_checker(self, images_u8):
        """device u8 [B,H,W,3] -> (images with flagged ones replaced by black, device int32 flags [B] | None)."""
        if self.safety_checker is None:
            return images_u8, None
        return self.safety_checker.forward(images_u8)

    # ---- pieces ------------------------------------------------------------------------
    def _need_device(self):
        if self.unet is None:
            raise RuntimeError("pipeline not placed on a device: call .to('cuda:0', dtype) first")
        if torch.cuda.current_device() != self.device.index:       # another pipeline / caller switched devices since
            torch.cuda.set_device(self.device)

    def encode_prompts(self, ids):
        """ids: int array/tensor [n,77] -> [n,77,ctx_dim] device tensor (CLIP text tower)."""
        self._need_device()
        if not torch.is_tensor(ids):
            ids = torch.as_tensor(np.asarray(ids))
        return self.text_encoder.forward(ops.h2d(ids, self.device))

    def _negative_context(self, neg_ids):
        key = np.asarray(neg_ids).tobytes()
        if key not in self._neg_cache:           # constant for a whole run -> encode once
            self._neg_cache[key] = self.encode_prompts(neg_ids)
        return self._neg_cache[key]

    def latents_to_device(self, latents):
        """[B,4,h,w] noise (any float dtype, CPU) -> channels-last [B,h,w,8] in compute dtype."""
        x = latents.to(torch.float32).permute(0, 2, 3, 1)
        x = torch.nn.functional.pad(x, (0, 8 - x.shape[-1]))
        return ops.h2d((x * self.scheduler.init_noise_sigma).contiguous(), self.device, self.dtype).contiguous()

    def _positive_context(self, prompt_ids, query_embeds=None):
        return self.encode_prompts(prompt_ids)

    def _sample(self, x2, b, hw, ctx, cemb2, steps, guidance_scale, cscale, t_start=0):
        """The denoising loop on the CFG-doubled latents x2 [2B,h,w,8] (in place).  t_start > 0 (img2img / SDEdit): only the
        timesteps from index t_start of the `steps`-step schedule run (DDIM only)."""
        sch = self.scheduler
        nc = self.cfgs["unet"]["out_channels"]
        if t_start and isinstance(sch, PNDMScheduler):
            raise NotImplementedError("img2img runs on DDIM / UniPC (the reference switches non-BLIP pipelines away from PNDM)")
        if graphs_enabled():
            if isinstance(sch, PNDMScheduler):
                plan = sch.plan(steps)                 # N+1 evaluations, the second timestep twice
                ts, plan = [t for t, _ in plan], [d for _, d in plan]
            elif isinstance(sch, UniPCMultistepScheduler):
                if t_start:
                    raise NotImplementedError("img2img with UniPC: the multistep history would have to start mid-schedule")
                plan = sch.plan(steps)
                ts, plan = [t for t, _ in plan], [r for _, r in plan]
            else:
                ts, plan = list(sch.set_timesteps(steps))[t_start:], None
            x2.copy_(self._step_graph(x2, cemb2, ctx, len(ts) if t_start else steps, True, guidance_scale, cscale, ts).run_on(x2, cemb2, ctx, ts, plan=plan))
            return
        nets = [self.unet] + ([self.controlnet] if self.controlnet is not None else [])
        for net in nets:
            net.prepare_context(ctx)
        eps = torch.zeros_like(x2)

        rec = ops._RECORDER
        twin_rec = rec is not None and getattr(rec, "twin", False) and self.controlnet is not None and fork_enabled()
        gate = torch.zeros(2, dtype=torch.int64, device=x2.device) if twin_rec else None

        def evaluate(i):
            if self.controlnet is None:
                mid, skips = self.unet.encode(x2, i)
            elif twin_rec:
                # a launch recorder that wants the TIMED path's dispatch (bench.Recorder(twin=True)): the two encoders on two streams
                # with the shared-chip hint, as the captured step runs them.  Both streams are held behind one sleeping wave
                # (ops.clock_probe, ~40 ms) until the host has enqueued both launch sequences, so that they really run side by side
                main, side = torch.cuda.current_stream(), side_stream(x2.device)
                ops.clock_probe(gate, 10000)
                side.wait_stream(main)
                rec.begin_twin()
                with ops.twin_branch(True):
                    with torch.cuda.stream(side):
                        cmid, cfeats = self.controlnet.encode(x2, i, conv_in_residual=cemb2)
                    mid, skips = self.unet.encode(x2, i)
                rec.end_twin()
                main.wait_stream(side)
                for t in (cmid, *cfeats):
                    t.record_stream(main)
                skips, mid = self.controlnet.zero_convs(cmid, cfeats, cscale, skips, mid)
            else:
                # the same launches, with the same dispatch decisions, as the two-branch graph step (_StepGraph._step): the
                # split-K of the paired encoders is sized for two concurrent branches (ops.twin_branch) -- except under a
                # launch recorder, which times every launch ALONE and therefore gets the full-chip dispatch
                with ops.twin_branch(fork_enabled() and ops._RECORDER is None):
                    cmid, cfeats = self.controlnet.encode(x2, i, conv_in_residual=cemb2)
                    mid, skips = self.unet.encode(x2, i)
                skips, mid = self.controlnet.zero_convs(cmid, cfeats, cscale, skips, mid)
            self.unet.decode(mid, skips, i, out=eps)

        if isinstance(sch, PNDMScheduler):
            plan = sch.plan(steps)                     # N+1 evaluations, the second timestep twice
            ts = [t for t, _ in plan]
            for net in nets:
                net.prepare_timesteps(ts)
            hist = torch.zeros((4,) + tuple(x2[:b].shape), device=x2.device, dtype=x2.dtype)
            saved = None
            for i, (t, d) in enumerate(plan):
                if d["save_sample"]:
                    saved = x2[:b].clone()
                evaluate(i)
                ops.cfg_plms_step(eps, x2, hist, saved if d["use_saved"] else None, b, hw, nc, guidance_scale,
                                  d["store_slot"], d["w_cur"], d["w_hist"], d["coef_sample"], d["coef_model"])
        elif isinstance(sch, UniPCMultistepScheduler):
            if t_start:
                raise NotImplementedError("img2img with UniPC: the multistep history would have to start mid-schedule")
            plan = sch.plan(steps)
            ts = [t for t, _ in plan]
            for net in nets:
                net.prepare_timesteps(ts)
            state = torch.zeros((3,) + tuple(x2[:b].shape), device=x2.device, dtype=x2.dtype)
            for i, (t, row) in enumerate(plan):
                evaluate(i)
                ops.cfg_unipc_step(eps, x2, state, b, hw, nc, guidance_scale, row=row)
        else:
            ts = list(sch.set_timesteps(steps))[t_start:]
            for net in nets:
                net.prepare_timesteps(ts)
            for i, t in enumerate(ts):
                evaluate(i)
                ops.cfg_ddim_step(eps, x2, b, hw, nc, guidance_scale, *sch.step_coefficients(t))

    def _step_graph(self, x, cemb, ctx, steps, cfg, guidance, cscale, ts):
        # the timesteps are part of the key: the cached time tables / coefficients follow the scheduler's configuration
        key = (tuple(x.shape), tuple(cemb.shape), tuple(ctx.shape), int(steps), bool(cfg), float(guidance), float(cscale), x.dtype,
               type(self.scheduler).__name__, tuple(int(t) for t in ts))
        g = self._graphs.pop(key, None)
        if g is None:
            while len(self._graphs) >= 3:                   # each graph keeps one step's activations resident
                self._graphs.pop(next(iter(self._graphs)))
            g = _StepGraph(self, x.shape, cemb.shape, steps, cfg, guidance, cscale)
        self._graphs[key] = g                               # most recently used last
        return g

    @torch.no_grad()
    def generate_batch(self, prompt_ids, negative_ids, control_u8, latents, num_inference_steps,
                       guidance_scale=7.5, controlnet_conditioning_scale=0.75, return_latents=False,
                       latents_on_device=False, query_embeds=None):
        """prompt_ids [B,77], negative_ids [1,77] or [B,77], control_u8 u8 [B,H,W,3] (numpy or
        device tensor), latents [B,4,H/8,W/8] noise (or, with latents_on_device, the
        channels-last [B,H/8,W/8,8] device tensor from `latents_to_device`).
        Returns a device u8 tensor [B,H,W,3]."""
        self._need_device()
        if guidance_scale <= 1.0:
            raise NotImplementedError("guidance_scale <= 1 (no CFG) belongs to the SDXL-Turbo branch (SURVEY a9)")
        dev, dt = self.device, self.dtype
        # (np.array: a writable copy -- arrays that come out of PIL are read-only and torch warns about wrapping those)
        ctrl = torch.from_numpy(np.array(control_u8)) if not torch.is_tensor(control_u8) else control_u8
        ctrl = ops.h2d(ctrl, dev).contiguous()
        b, hh, ww, _ = ctrl.shape
        mult = 8 << (len(self.cfgs["unet"]["block_out"]) - 1)
        if hh % mult or ww % mult:
            # the UNet halves the latent (H/8 x W/8) once per level below the first and doubles it back exactly; the
            # reference only ever feeds multiples of 64 (all_utils/utils.py:65-77 rounds both sides to 64)
            raise ValueError(f"control image sides must be multiples of {mult} (got {hh}x{ww})")
        want = (b, hh // 8, ww // 8, 8) if latents_on_device else (b, self.cfgs["unet"]["in_channels"], hh // 8, ww // 8)
        if tuple(latents.shape) != want:
            raise ValueError(f"latents shape {tuple(latents.shape)} does not match the control image {hh}x{ww}")
        pos = self._positive_context(prompt_ids, query_embeds)
        neg = self._negative_context(negative_ids)
        if neg.shape[0] == 1 and b > 1:
            neg = neg.expand(b, -1, -1)
        if neg.shape[1] != pos.shape[1]:
            raise ValueError(f"negative context has {neg.shape[1]} tokens, positive {pos.shape[1]}")
        ctx = torch.cat([neg, pos], 0).contiguous()                       # [2B,77,C]: uncond first
        cond = ops.u8_to_act(ctrl, dt)
        cemb = self.controlnet.cond_embedding(cond)
        cemb2 = torch.cat([cemb, cemb], 0)
        x = latents if latents_on_device else self.latents_to_device(latents)
        x2 = torch.cat([x, x], 0).contiguous()
        h8, w8 = hh // 8, ww // 8
        self._sample(x2, b, h8 * w8, ctx, cemb2, num_inference_steps, guidance_scale, controlnet_conditioning_scale)
        z = ops.scale(x2[:b], 1.0 / self.cfgs["vae"]["scaling_factor"])
        img = self.vae.decode(z)
        out, self.last_nsfw = self.run_safety_checker(ops.act_to_u8(img))
        if return_latents:
            return out, x2[:b], img
        return out

    # ---- the reference's call form ----------------------------------------------------
    def __call__(self, prompt=None, image=None, num_inference_steps=50, generator=None, guidance_scale=7.5,
                 negative_prompt=None, controlnet_conditioning_scale=1.0, **unused):
        self._need_device()
        if image is None or prompt is None:
            raise ValueError("`prompt` and `image` (the control image) are required")
        ctrl = np.asarray(image.convert("RGB") if isinstance(image, Image.Image) else image, dtype=np.uint8)
        if ctrl.ndim != 3 or ctrl.shape[2] != 3:
            raise ValueError("control image must be RGB")
        hh, ww = ctrl.shape[:2]
        if generator is not None and generator.device.type != "cpu":
            raise NotImplementedError("the reference passes the global CPU generator (run_aug/run_aug.py:324)")
        lat = torch.randn((1, self.cfgs["unet"]["in_channels"], hh // 8, ww // 8), generator=generator,
                          dtype=self.noise_dtype)
        ids = self.tokenizer(str(prompt))
        neg = self.tokenizer(negative_prompt if negative_prompt is not None else "")
        out = self.generate_batch(ids, neg, ctrl[None], lat, num_inference_steps, guidance_scale,
                                  controlnet_conditioning_scale)
        arr = out.cpu().numpy()
        nsfw = None if self.last_nsfw is None else [bool(v) for v in self.last_nsfw.cpu().tolist()]
        return PipelineOutput([Image.fromarray(a) for a in arr], nsfw)


class StableDiffusionControlNetImg2ImgPipeline(StableDiffusionControlNetPipeline):
    """Drop-in for diffusers' `StableDiffusionControlNetImg2ImgPipeline` as the reference builds it with SDEDIT = 1
    (run_aug/run_aug.py:203-206) and calls it (:252-260, :274-276): `pipe(prompt, image=<source PIL>, control_image=<canny
    PIL>, strength, num_inference_steps, generator, guidance_scale, negative_prompt, controlnet_conditioning_scale)`.
    The source image is encoded by the VAE encoder (HIP launch graph), the latent is sampled from the posterior and noised
    to the first kept timestep in one kernel (`saspa_vae_sample_noise`; the two noise draws come from the CPU generator in
    the reference's order: posterior sample first, then the scheduler noise), and the last int(steps * strength) DDIM steps
    run exactly like the text-to-image loop (same captured step graph)."""

    def _build_extra(self, sd, cf, device, cdt):
        super()._build_extra(sd, cf, device, cdt)
        if "encoder.conv_in.weight" not in sd["vae"]:
            raise KeyError("the VAE checkpoint has no encoder half: img2img (SDEdit) needs vae/encoder.* and quant_conv")
        self.vae_encoder = models.VAEEncoder(sd["vae"], cf["vae"], device, cdt)

    @staticmethod
    def kept_steps(num_inference_steps, strength):
        """get_timesteps: (index of the first kept timestep, number of kept steps)."""
        init = min(int(num_inference_steps * strength), num_inference_steps)
        t_start = max(num_inference_steps - init, 0)
        return t_start, num_inference_steps - t_start

    @torch.no_grad()
    def generate_batch_img2img(self, prompt_ids, negative_ids, source_u8, control_u8, sample_noise, noise, num_inference_steps,
                               strength, guidance_scale=7.5, controlnet_conditioning_scale=0.75, return_latents=False):
        """source_u8 / control_u8: u8 [B,H,W,3] (numpy or device); sample_noise / noise: [B,4,H/8,W/8] CPU draws."""
        self._need_device()
        if guidance_scale <= 1.0:
            raise NotImplementedError("guidance_scale <= 1 (no CFG) belongs to the SDXL-Turbo branch")
        if not 0.0 <= strength <= 1.0:
            raise ValueError(f"The value of strength should in [0.0, 1.0] but is {strength}")
        t_start, kept = self.kept_steps(num_inference_steps, strength)
        if kept < 1:
            raise ValueError(f"After adjusting the num_inference_steps by strength parameter: {strength}, the number of pipeline "
                             f"steps is {kept} which is < 1 and not appropriate for this pipeline.")
        dev, dt = self.device, self.dtype
        to_dev = lambda a: ops.h2d(a if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)), dev).contiguous()   # noqa: E731
        if (control_u8 is None) != (self.controlnet is None):
            raise ValueError("a control image is required by the ControlNet pipelines and refused by the ControlNet-free one")
        src = to_dev(source_u8)
        ctrl = to_dev(control_u8) if control_u8 is not None else src
        b, hh, ww, _ = ctrl.shape
        if tuple(src.shape) != tuple(ctrl.shape):
            raise ValueError("source and control images must have the same size")
        mult = 8 << (len(self.cfgs["unet"]["block_out"]) - 1)
        if hh % mult or ww % mult:
            raise ValueError(f"image sides must be multiples of {mult} (got {hh}x{ww})")
        from . import imageproc
        px = imageproc.normalize_u8(src, dt, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))          # VaeImageProcessor: [0,255] -> [-1,1]
        moments = self.vae_encoder.encode(px)
        sch = self.scheduler
        ts = list(sch.set_timesteps(num_inference_steps))
        a_t = float(sch.alphas_cumprod[int(ts[t_start])])
        x = ops.vae_sample_noise(moments, self.latents_to_device(sample_noise), self.latents_to_device(noise),
                                 self.cfgs["vae"]["scaling_factor"], a_t ** 0.5, (1.0 - a_t) ** 0.5)
        pos = self._positive_context(prompt_ids)
        neg = self._negative_context(negative_ids)
        if neg.shape[0] == 1 and b > 1:
            neg = neg.expand(b, -1, -1)
        ctx = torch.cat([neg, pos], 0).contiguous()
        if self.controlnet is not None:
            cemb = self.controlnet.cond_embedding(ops.u8_to_act(ctrl, dt))
            cemb2 = torch.cat([cemb, cemb], 0)
        else:
            cemb2 = torch.zeros((1,), device=dev, dtype=dt)           # placeholder: nothing reads it
        x2 = torch.cat([x, x], 0).contiguous()
        self._sample(x2, b, (hh // 8) * (ww // 8), ctx, cemb2, num_inference_steps, guidance_scale,
                     controlnet_conditioning_scale, t_start=t_start)
        img = self.vae.decode(ops.scale(x2[:b], 1.0 / self.cfgs["vae"]["scaling_factor"]))
        out, self.last_nsfw = self.run_safety_checker(ops.act_to_u8(img))
        if return_latents:
            return out, x2[:b], img
        return out

    def __call__(self, prompt=None, image=None, control_image=None, strength=0.8, num_inference_steps=50, generator=None,
                 guidance_scale=7.5, negative_prompt=None, controlnet_conditioning_scale=0.8, **unused):
        self._need_device()
        if image is None or control_image is None or prompt is None:
            raise ValueError("`prompt`, `image` (the source image) and `control_image` are required")
        as_u8 = lambda im: np.asarray(im.convert("RGB") if isinstance(im, Image.Image) else im, dtype=np.uint8)   # noqa: E731
        src, ctrl = as_u8(image), as_u8(control_image)
        hh, ww = ctrl.shape[:2]
        if generator is not None and generator.device.type != "cpu":
            raise NotImplementedError("the reference passes the global CPU generator (run_aug/run_aug.py:324)")
        shape = (1, self.cfgs["unet"]["in_channels"], hh // 8, ww // 8)
        e1 = torch.randn(shape, generator=generator, dtype=self.noise_dtype)             # latent_dist.sample(generator)
        e2 = torch.randn(shape, generator=generator, dtype=self.noise_dtype)             # randn_tensor for add_noise
        ids = self.tokenizer(str(prompt))
        neg = self.tokenizer(negative_prompt if negative_prompt is not None else "")
        out = self.generate_batch_img2img(ids, neg, src[None], ctrl[None], e1, e2, num_inference_steps, strength, guidance_scale,
                                          controlnet_conditioning_scale)
        nsfw = None if self.last_nsfw is None else [bool(v) for v in self.last_nsfw.cpu().tolist()]
        return PipelineOutput([Image.fromarray(a) for a in out.cpu().numpy()], nsfw)


class StableDiffusionImg2ImgPipeline(StableDiffusionControlNetImg2ImgPipeline):
    """Drop-in for diffusers' `StableDiffusionImg2ImgPipeline` -- the reference's CONTROLNET = None, SDEDIT = 1 branch
    (run_aug/run_aug.py:163-165; the Real-Guidance baseline, defaults in run_aug/run_aug_real_guidance.py:520-523), called
    as `pipe(prompt, image=<source PIL>, strength, num_inference_steps, generator, guidance_scale, negative_prompt)`
    (:235-241, :274-276).  Same VAE-encode / posterior sample / add-noise / last int(steps * strength) DDIM steps as the
    ControlNet img2img pipeline; every evaluation is the UNet alone (no ControlNet is built, no control image exists)."""
    HAS_CONTROLNET = False

    @classmethod
    def from_pretrained(cls, base_dir, cfgs=SD15):
        def f(d, *names):
            for n in names:
                p = os.path.join(d, n)
                if os.path.exists(p):
                    return W.load_safetensors(p)
            raise FileNotFoundError(f"no safetensors weights in {d}")
        w = ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors")
        sds = dict(unet=f(os.path.join(base_dir, "unet"), *w), vae=f(os.path.join(base_dir, "vae"), *w),
                   text=f(os.path.join(base_dir, "text_encoder"), "model.safetensors", "model.fp16.safetensors"))
        sc = os.path.join(base_dir, "safety_checker")
        if "safety" in cfgs and os.path.isdir(sc):
            sds["safety"] = f(sc, "model.safetensors", "model.fp16.safetensors")
        return cls(sds, cfgs, tokenizer=make_tokenizer(os.path.join(base_dir, "tokenizer"), cfgs["text"]["vocab"]))

    def generate_batch_img2img(self, prompt_ids, negative_ids, source_u8, sample_noise, noise, num_inference_steps, strength,
                               guidance_scale=7.5, return_latents=False):
        """source_u8: u8 [B,H,W,3] (numpy or device); sample_noise / noise: [B,4,H/8,W/8] CPU draws (posterior, add_noise)."""
        return super().generate_batch_img2img(prompt_ids, negative_ids, source_u8, None, sample_noise, noise, num_inference_steps,
                                              strength, guidance_scale, 0.0, return_latents)

    def __call__(self, prompt=None, image=None, strength=0.8, num_inference_steps=50, generator=None, guidance_scale=7.5,
                 negative_prompt=None, **unused):
        self._need_device()
        if image is None or prompt is None:
            raise ValueError("`prompt` and `image` (the source image) are required")
        src = np.asarray(image.convert("RGB") if isinstance(image, Image.Image) else image, dtype=np.uint8)
        hh, ww = src.shape[:2]
        if generator is not None and generator.device.type != "cpu":
            raise NotImplementedError("the reference passes the global CPU generator (run_aug/run_aug.py:324)")
        shape = (1, self.cfgs["unet"]["in_channels"], hh // 8, ww // 8)
        e1 = torch.randn(shape, generator=generator, dtype=self.noise_dtype)             # latent_dist.sample(generator)
        e2 = torch.randn(shape, generator=generator, dtype=self.noise_dtype)             # randn_tensor for add_noise
        ids = self.tokenizer(str(prompt))
        neg = self.tokenizer(negative_prompt if negative_prompt is not None else "")
        out = self.generate_batch_img2img(ids, neg, src[None], e1, e2, num_inference_steps, strength, guidance_scale)
        nsfw = None if self.last_nsfw is None else [bool(v) for v in self.last_nsfw.cpu().tolist()]
        return PipelineOutput([Image.fromarray(a) for a in out.cpu().numpy()], nsfw)


class BlipDiffusionControlNetPipeline(StableDiffusionControlNetPipeline):
    """Drop-in for diffusers' `BlipDiffusionControlNetPipeline` as the reference builds and calls it for every dataset
    but planes (run_aug/run_aug.py:181, :211, :243-250, :262-265, :521):

        pipe = BlipDiffusionControlNetPipeline.from_pretrained("Salesforce/blipdiffusion-controlnet").to(DEVICE, fp16)
        image = pipe(prompt=..., reference_image=<PIL same-class image>, condtioning_image=<canny PIL>,
                     source_subject_category="bird", target_subject_category="bird", height=H, width=W,
                     neg_prompt=..., num_inference_steps=..., generator=..., guidance_scale=7.5).images[0]

    Differences from the SD-1.5 pipeline, all mirrored: the PNDM / PLMS scheduler of the checkpoint is kept (N+1
    network evaluations), no ControlNet conditioning scale is passed (= 1.0), the prompt is rewritten
    "a {category} {prompt}" and repeated prompt_strength * prompt_reps (= 20) times, tokenised to 77 - 16 tokens,
    and the 16 subject tokens of the Q-Former front-end (saspa_aug_amd.blip.Blip2QFormer) are spliced into the
    CLIP token embeddings at position 2.  UNet / ControlNet / VAE are the SD-1.5 architectures (shared kernels)."""

    def __init__(self, state_dicts, cfgs=BLIP_DIFFUSION, tokenizer=None, scheduler=None, qformer_tokenizer=None):
        super().__init__(state_dicts, cfgs, tokenizer, scheduler or PNDMScheduler())
        self.qformer_tokenizer = qformer_tokenizer or make_bert_tokenizer(vocab=cfgs["qformer"]["vocab"])
        self.qformer = None

    @classmethod
    def from_synthetic(cls, cfgs=BLIP_DIFFUSION, seed=0):
        return cls(W.synth_family(cfgs, seed), cfgs)

    @classmethod
    def from_pretrained(cls, repo_dir, cfgs=BLIP_DIFFUSION):
        """Local copy of Salesforce/blipdiffusion-controlnet: unet/ vae/ text_encoder/ controlnet/ qformer/ tokenizer/."""
        def f(sub, *names):
            for n in names:
                p = os.path.join(repo_dir, sub, n)
                if os.path.exists(p):
                    return W.load_safetensors(p)
            raise FileNotFoundError(f"no safetensors weights in {os.path.join(repo_dir, sub)}")
        names = ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors", "model.safetensors")
        sds = dict(unet=f("unet", *names), vae=f("vae", *names), text=f("text_encoder", *names),
                   controlnet=f("controlnet", *names), qformer=f("qformer", *names))
        return cls(sds, cfgs, tokenizer=make_tokenizer(os.path.join(repo_dir, "tokenizer"), cfgs["text"]["vocab"]),
                   qformer_tokenizer=make_bert_tokenizer(os.path.join(repo_dir, "qformer"), cfgs["qformer"]["vocab"]))

    def _build_extra(self, sd, cf, device, cdt):
        from .blip import Blip2QFormer
        self.qformer = Blip2QFormer(sd["qformer"], cf["qformer"], device, cdt)      # (this pipeline has no safety checker)

    # ---- pieces ------------------------------------------------------------------------
    @staticmethod
    def build_prompt(prompt, tgt_subject, prompt_strength=1.0, prompt_reps=20):
        p = f"a {tgt_subject} {str(prompt).strip()}"
        return ", ".join([p] * int(prompt_strength * prompt_reps))

    def prompt_token_count(self):
        return self.cfgs["text"]["max_pos"] - self.cfgs["qformer"]["num_query"]

    def get_query_embeddings(self, reference_images, source_subject_categories):
        """reference images (PIL / u8 HWC) + category strings -> [B, 16, width] subject tokens (device)."""
        self._need_device()
        from .blip import preprocess_reference
        qc = self.cfgs["qformer"]
        def one(im):
            if torch.is_tensor(im) and im.is_cuda:          # device u8 [H,W,3] (run_aug): Pillow-exact bicubic on the device
                from . import imageproc
                x = imageproc.resize_u8(im[None].contiguous(), qc["image_size"], qc["image_size"], filt="bicubic")
                return imageproc.normalize_u8(x, torch.float32, BLIP_IMAGE_MEAN, BLIP_IMAGE_STD)[0, :, :, :3].permute(2, 0, 1)
            return preprocess_reference(im, qc, BLIP_IMAGE_MEAN, BLIP_IMAGE_STD).to(self.device)
        px = torch.stack([one(im) for im in reference_images])
        ids = [self.qformer_tokenizer(c) for c in source_subject_categories]
        if len({i.shape[1] for i in ids}) == 1:
            return self.qformer.forward(px, np.concatenate(ids))
        # categories of different token counts: no padding mask in the kernels -> one item at a time
        return torch.cat([self.qformer.forward(px[i:i + 1], ids[i]) for i in range(len(ids))], 0)

    def _positive_context(self, prompt_ids, query_embeds=None):
        if query_embeds is None:
            raise ValueError("BLIP-Diffusion needs the subject tokens (query_embeds); see get_query_embeddings")
        ids = torch.as_tensor(np.asarray(prompt_ids)) if not torch.is_tensor(prompt_ids) else prompt_ids
        if ids.shape[1] != self.prompt_token_count():
            raise ValueError(f"prompt must be tokenised to {self.prompt_token_count()} tokens (77 - subject tokens)")
        return self.text_encoder.forward(ops.h2d(ids, self.device), query_embeds, self.cfgs["ctx_begin_pos"])

    # ---- the reference's call form (keyword names as diffusers spells them, typo included) ----
    def __call__(self, prompt=None, reference_image=None, condtioning_image=None, source_subject_category=None,
                 target_subject_category=None, height=512, width=512, neg_prompt="", num_inference_steps=50,
                 generator=None, guidance_scale=7.5, prompt_strength=1.0, prompt_reps=20, **unused):
        self._need_device()
        if prompt is None or reference_image is None or condtioning_image is None:
            raise ValueError("`prompt`, `reference_image` and `condtioning_image` are required")
        ctrl = condtioning_image.convert("RGB") if isinstance(condtioning_image, Image.Image) else Image.fromarray(np.asarray(condtioning_image))
        if ctrl.size != (width, height):        # prepare_control_image resizes the control image to (width, height)
            ctrl = ctrl.resize((width, height), resample=Image.LANCZOS)
        ctrl = np.asarray(ctrl, dtype=np.uint8)
        if generator is not None and generator.device.type != "cpu":
            raise NotImplementedError("the reference passes the global CPU generator (run_aug/run_aug.py:324)")
        lat = torch.randn((1, self.cfgs["unet"]["in_channels"], height // 8, width // 8), generator=generator, dtype=self.noise_dtype)
        text = self.build_prompt(prompt, target_subject_category, prompt_strength, prompt_reps)
        ids = self.tokenizer(text, max_len=self.prompt_token_count())
        neg = self.tokenizer(neg_prompt or "")
        q = self.get_query_embeddings([reference_image], [source_subject_category])
        out = self.generate_batch(ids, neg, ctrl[None], lat, num_inference_steps, guidance_scale, 1.0, query_embeds=q)
        arr = out.cpu().numpy()
        return PipelineOutput([Image.fromarray(a) for a in arr], None)


class StableDiffusionXLControlNetPipeline(StableDiffusionControlNetPipeline):
    """Drop-in for diffusers' `StableDiffusionXLControlNetPipeline` as the reference builds and calls it for
    `BASE_MODEL = "sd_xl-turbo"` (its choice for CUB; run_aug/run_aug.py:189-201, :223-228, :564-571):

        vae  = AutoencoderKL.from_pretrained("madebyollin/sdxl-vae-fp16-fix")
        pipe = StableDiffusionXLControlNetPipeline.from_pretrained("stabilityai/sdxl-turbo", controlnet=<canny-sdxl>, vae=vae)
        pipe.scheduler = DDIMScheduler.from_config(pipe.scheduler.config); pipe.upcast_vae()
        image = pipe(prompt=..., image=<canny PIL>, num_inference_steps=2, generator=..., guidance_scale=0,
                     negative_prompt=None, controlnet_conditioning_scale=0.75).images[0]

    Differences from the SD-1.5 pipeline, all mirrored: guidance_scale <= 1 -> NO classifier-free guidance (one
    conditional evaluation per step, the negative prompt is never encoded); two text towers read at
    hidden_states[-2] and concatenated to a 2048-wide context, the second tower's projected EOS state is the pooled
    embedding of the `text_time` conditioning together with add_time_ids = (H, W, 0, 0, H, W); three-level UNet /
    ControlNet with transformer depths (2, 10) and linear projections; VAE scaling factor 0.13025, decoded in fp32
    after `upcast_vae()` (the exact-fp32 MFMA kernels) even when the denoiser runs in bf16."""

    def __init__(self, state_dicts, cfgs=SDXL_TURBO, tokenizer=None, scheduler=None, tokenizer_2=None):
        super().__init__(state_dicts, cfgs, tokenizer, scheduler or DDIMScheduler(**SDXL_TURBO_SCHEDULER_CONFIG))
        self.tokenizer_2 = tokenizer_2 or make_tokenizer(vocab=cfgs["text2"]["vocab"])
        self.text_encoder_2 = None
        self._vae_fp32 = False
        self._vae_gemm = None
        self._vae_sd = None

    @classmethod
    def from_synthetic(cls, cfgs=SDXL_TURBO, seed=0):
        return cls(W.synth_family(cfgs, seed), cfgs)

    @classmethod
    def from_pretrained(cls, base_dir, controlnet_dir, vae_dir=None, cfgs=SDXL_TURBO):
        """Local copies of stabilityai/sdxl-turbo (unet/ text_encoder/ text_encoder_2/ tokenizer/ tokenizer_2/ vae/),
        diffusers/controlnet-canny-sdxl-1.0 and (optionally) madebyollin/sdxl-vae-fp16-fix."""
        names = ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors", "model.safetensors",
                 "model.fp16.safetensors")

        def f(d):
            for n in names:
                p = os.path.join(d, n)
                if os.path.exists(p):
                    return W.load_safetensors(p)
            raise FileNotFoundError(f"no safetensors weights in {d}")
        sds = dict(unet=f(os.path.join(base_dir, "unet")), vae=f(vae_dir or os.path.join(base_dir, "vae")),
                   text=f(os.path.join(base_dir, "text_encoder")), text2=f(os.path.join(base_dir, "text_encoder_2")),
                   controlnet=f(controlnet_dir))
        return cls(sds, cfgs, tokenizer=make_tokenizer(os.path.join(base_dir, "tokenizer"), cfgs["text"]["vocab"]),
                   tokenizer_2=make_tokenizer(os.path.join(base_dir, "tokenizer_2"), cfgs["text2"]["vocab"]))

    def to(self, device, dtype=None):
        self._vae_sd = None if self._state_dicts is None else self._state_dicts["vae"]
        super().to(device, dtype)
        if self._vae_fp32:
            self.upcast_vae(self._vae_gemm)
        return self

    def _build_extra(self, sd, cf, device, cdt):
        self.text_encoder_2 = models.CLIPText(sd["text2"], cf["text2"], device, cdt)

    def upcast_vae(self, gemm=None):
        """run_aug/run_aug.py:224: the VAE decodes in float32.  Before `.to()` it is recorded; after, the decoder is
        re-packed for fp32 storage from the retained VAE state dict.  The reference upcasts because fp16 overflows in this
        VAE, not for the last mantissa bits: the decoder's GEMMs default to `SASPA_F32X3` (three bf16 MFMAs per product,
        ~1e-5 relative, well inside the 1e-3 per-pixel bar, several times the fp32 MFMA rate); `gemm="exact"` or
        SASPA_VAE_EXACT_FP32=1 selects the exact fp32 MFMA path.  GroupNorm / softmax / residuals are fp32 either way."""
        self._vae_fp32 = True
        if gemm is None:
            gemm = "exact" if os.environ.get("SASPA_VAE_EXACT_FP32", "0") == "1" else "x3"
        self._vae_gemm = gemm
        if self.vae is not None and (self.vae.dtype != torch.float32 or self.vae.f32_gemm != gemm):
            self.vae = models.VAEDecoder(self._vae_sd, self.cfgs["vae"], self.device, torch.float32, f32_gemm=gemm)
        return self

    # ---- pieces ------------------------------------------------------------------------
    def pad_ids_2(self, ids):
        """tokenizer_2 pads with id 0 ("!") after the first EOS instead of repeating EOS."""
        ids = np.array(ids, np.int64, copy=True)
        eos = ids.max(axis=1, keepdims=True)
        first = (ids == eos).argmax(axis=1)
        for r in range(ids.shape[0]):
            ids[r, first[r] + 1:] = self.cfgs["text2"].get("pad_id", 0)
        return ids

    def encode_prompts_xl(self, ids1, ids2):
        """-> (context [B,77,ctx1+ctx2], pooled [B, proj] fp32)."""
        self._need_device()
        t = lambda a: ops.h2d(a if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)), self.device)   # noqa: E731
        h1, _ = self.text_encoder.forward(t(ids1), penultimate=True)
        h2, pooled = self.text_encoder_2.forward(t(ids2), penultimate=True)
        return torch.cat([h1, h2], -1).contiguous(), pooled.float()

    @torch.no_grad()
    def generate_batch(self, prompt_ids, negative_ids, control_u8, latents, num_inference_steps, guidance_scale=0.0,
                       controlnet_conditioning_scale=0.75, return_latents=False, latents_on_device=False,
                       prompt_ids_2=None, negative_ids_2=None, **unused):
        """Same contract as the SD-1.5 form.  guidance_scale <= 1 (the reference's sd_xl-turbo operating point,
        run_aug/run_aug.py:568): no CFG, `negative_ids` ignored.  guidance_scale > 1: classifier-free guidance with the
        negative prompt (None -> zero embeddings).  `prompt_ids_2` / `negative_ids_2` default to the tokenizer_2 padding
        of the first tokenizer's ids."""
        self._need_device()
        cfg = guidance_scale > 1.0                         # diffusers: do_classifier_free_guidance = guidance_scale > 1
        dev, dt = self.device, self.dtype
        # (np.array: a writable copy -- arrays that come out of PIL are read-only and torch warns about wrapping those)
        ctrl = torch.from_numpy(np.array(control_u8)) if not torch.is_tensor(control_u8) else control_u8
        ctrl = ops.h2d(ctrl, dev).contiguous()
        b, hh, ww, _ = ctrl.shape
        mult = 8 << (len(self.cfgs["unet"]["block_out"]) - 1)
        if hh % mult or ww % mult:
            # the UNet halves the latent (H/8 x W/8) once per level below the first and doubles it back exactly; the
            # reference only ever feeds multiples of 64 (all_utils/utils.py:65-77 rounds both sides to 64)
            raise ValueError(f"control image sides must be multiples of {mult} (got {hh}x{ww})")
        want = (b, hh // 8, ww // 8, 8) if latents_on_device else (b, self.cfgs["unet"]["in_channels"], hh // 8, ww // 8)
        if tuple(latents.shape) != want:
            raise ValueError(f"latents shape {tuple(latents.shape)} does not match the control image {hh}x{ww}")
        ids1 = np.asarray(prompt_ids.cpu() if torch.is_tensor(prompt_ids) else prompt_ids)
        ids2 = self.pad_ids_2(ids1) if prompt_ids_2 is None else prompt_ids_2
        ctx, pooled = self.encode_prompts_xl(ids1, ids2)
        if cfg:
            # BASELINE configs[4] family (guidance on): uncond first.  negative_prompt None -> ZERO embeddings (sdxl-turbo's
            # model_index.json has force_zeros_for_empty_prompt = true), otherwise both towers encode the negative prompt
            if negative_ids is None:
                nctx, npooled = torch.zeros_like(ctx), torch.zeros_like(pooled)
            else:
                n1 = np.asarray(negative_ids.cpu() if torch.is_tensor(negative_ids) else negative_ids)
                n2 = self.pad_ids_2(n1) if negative_ids_2 is None else negative_ids_2
                nctx, npooled = self.encode_prompts_xl(n1, n2)
                if nctx.shape[0] == 1 and b > 1:
                    nctx, npooled = nctx.expand(b, -1, -1), npooled.expand(b, -1)
            ctx = torch.cat([nctx, ctx], 0).contiguous()
            pooled = torch.cat([npooled, pooled], 0).contiguous()
        nb = 2 * b if cfg else b
        time_ids = [[hh, ww, 0, 0, hh, ww]] * nb           # original_size + crops_coords_top_left + target_size
        cond = ops.u8_to_act(ctrl, dt)
        cemb = self.controlnet.cond_embedding(cond)
        x = (latents if latents_on_device else self.latents_to_device(latents)).contiguous()
        if cfg:
            x = torch.cat([x, x], 0).contiguous()
            cemb = torch.cat([cemb, cemb], 0)
        sch = self.scheduler
        unipc = isinstance(sch, UniPCMultistepScheduler)        # run_aug/run_aug.py:223-226 with sampler = "unipcmultistep"
        if unipc:
            plan = sch.plan(num_inference_steps)
            ts, rows = [t for t, _ in plan], [r for _, r in plan]
        else:
            ts, rows = sch.set_timesteps(num_inference_steps), None
        if graphs_enabled():
            g = self._step_graph(x, cemb, ctx, num_inference_steps, cfg, float(guidance_scale) if cfg else 0.0,
                                 controlnet_conditioning_scale, ts)
            x = g.run_on(x, cemb, ctx, ts, (pooled, time_ids), plan=rows).clone()
        else:
            for net in (self.unet, self.controlnet):
                net.prepare_context(ctx)
                net.prepare_timesteps(ts, (pooled, time_ids))
            eps = torch.zeros_like(x)
            nc, hw = self.cfgs["unet"]["out_channels"], (hh // 8) * (ww // 8)
            state = torch.zeros((3, b) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype) if unipc else None
            for i, t in enumerate(ts):
                # the same launches, with the same dispatch decisions, as the two-branch graph step (_StepGraph._step): the paired
                # encoders carry the shared-chip hint (ops.twin_branch) -- except under a launch recorder (every launch alone)
                with ops.twin_branch(fork_enabled() and ops._RECORDER is None):
                    cmid, cfeats = self.controlnet.encode(x, i, conv_in_residual=cemb)
                    mid, skips = self.unet.encode(x, i)
                skips2, mid2 = self.controlnet.zero_convs(cmid, cfeats, controlnet_conditioning_scale, skips, mid)
                self.unet.decode(mid2, skips2, i, out=eps)
                if unipc and cfg:
                    ops.cfg_unipc_step(eps, x, state, b, hw, nc, guidance_scale, row=rows[i])
                elif unipc:
                    ops.unipc_step(eps, x, state, b, hw, nc, row=rows[i])
                elif cfg:
                    ops.cfg_ddim_step(eps, x, b, hw, nc, guidance_scale, *sch.step_coefficients(t))
                else:
                    ops.ddim_step(eps, x, b, hw, nc, *sch.step_coefficients(t))
        x = x[:b].contiguous() if cfg else x
        z = ops.scale(x, 1.0 / self.cfgs["vae"]["scaling_factor"])
        if self.vae.dtype != z.dtype:
            z = z.to(self.vae.dtype)                      # upcast_vae(): latents follow the VAE dtype
        img = self.vae.decode(z)
        out = ops.act_to_u8(img)
        if return_latents:
            return out, x, img
        return out

    def __call__(self, prompt=None, image=None, num_inference_steps=50, generator=None, guidance_scale=5.0,
                 negative_prompt=None, controlnet_conditioning_scale=1.0, **unused):
        self._need_device()
        if image is None or prompt is None:
            raise ValueError("`prompt` and `image` (the control image) are required")
        ctrl = np.asarray(image.convert("RGB") if isinstance(image, Image.Image) else image, dtype=np.uint8)
        if ctrl.ndim != 3 or ctrl.shape[2] != 3:
            raise ValueError("control image must be RGB")
        hh, ww = ctrl.shape[:2]
        if generator is not None and generator.device.type != "cpu":
            raise NotImplementedError("the reference passes the global CPU generator (run_aug/run_aug.py:324)")
        lat = torch.randn((1, self.cfgs["unet"]["in_channels"], hh // 8, ww // 8), generator=generator, dtype=self.noise_dtype)
        ids1 = self.tokenizer(str(prompt))
        ids2 = self.pad_ids_2(self.tokenizer_2(str(prompt)))
        n1 = n2 = None
        if negative_prompt is not None and guidance_scale > 1.0:
            n1 = self.tokenizer(str(negative_prompt))
            n2 = self.pad_ids_2(self.tokenizer_2(str(negative_prompt)))
        out = self.generate_batch(ids1, n1, ctrl[None], lat, num_inference_steps, guidance_scale,
                                  controlnet_conditioning_scale, prompt_ids_2=ids2, negative_ids_2=n2)
        arr = out.cpu().numpy()
        return PipelineOutput([Image.fromarray(a) for a in arr], None)
