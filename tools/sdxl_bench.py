"""BASELINE.json configs[4] timing (parity-test configuration, not the bench line): SDXL-Turbo + Canny ControlNet at
full width, synthetic weights, bf16 denoiser + fp32-upcast VAE (run_aug/run_aug.py:224).  Two operating points:
the reference's own (512x512, 2 DDIM steps, no CFG, run_aug/run_aug.py:564-571) and the BASELINE stretch shape
(1024x1024, 4 steps).  usage: python tools/sdxl_bench.py [batch] [--bf16-vae | --exact-vae] [--fp8]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import saspa_aug_amd  # noqa: F401
from saspa_aug_amd import config as CFG, ops
from saspa_aug_amd.pipeline import StableDiffusionXLControlNetPipeline
from saspa_aug_amd.synthetic import synthetic_image, synthetic_prompt_ids
dev = torch.device('cuda:0')
b = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
t0 = time.time()
pipe = StableDiffusionXLControlNetPipeline.from_synthetic(CFG.SDXL_TURBO, 0)
if "--fp8" in sys.argv:
    pipe.enable_fp8()
if "--bf16-vae" not in sys.argv:
    pipe.upcast_vae("exact" if "--exact-vae" in sys.argv else None)      # default: fp32 storage, SASPA_F32X3 GEMMs
pipe = pipe.to(dev, torch.bfloat16)
print(f"built in {time.time() - t0:.0f} s", flush=True)
ids = synthetic_prompt_ids(b)
for res, steps in ((512, 2), (1024, 4)):
    imgs = torch.from_numpy(np.stack([synthetic_image(res, res, i) for i in range(b)])).to(dev)
    lat = torch.randn((b, 4, res // 8, res // 8), generator=torch.manual_seed(1), dtype=torch.float16)

    def once():
        ctrl = ops.canny(imgs, 120, 200)
        return pipe.generate_batch(ids, None, ctrl, lat, steps, 0.0, 0.75)

    out = once(); torch.cuda.synchronize()
    n = 3
    t1 = time.time()
    for _ in range(n): out = once()
    torch.cuda.synchronize(); dt = (time.time() - t1) / n
    out2 = once()
    # denoiser alone (no VAE): time one more run's sampling loop through the recorder-free path
    torch.cuda.synchronize(); t2 = time.time()
    z = torch.randn((b, res // 8, res // 8, 8), device=dev).to(pipe.vae.dtype)
    pipe.vae.decode(z); torch.cuda.synchronize(); tv = time.time() - t2
    # share of ONE UNet + ControlNet evaluation's algorithmic FLOPs (implicit GEMMs + attention products) that ran on fp8 tiles: every
    # recorded launch carries its kernel (bench.Recorder / ops._meta_kernel; family 8 = saspa_gemm_fp8); a generation with 2 x steps
    # minus one with steps leaves exactly `steps` evaluations (text towers, conditioning embedding, VAE cancel)
    from bench import Recorder

    def recorded(nsteps):
        rec = Recorder()
        ops.set_recorder(rec)
        pipe.generate_batch(ids, None, ops.canny(imgs, 120, 200), lat, nsteps, 0.0, 0.75)
        ops.set_recorder(None)
        torch.cuda.synchronize()
        f8 = sum(fl for k, fl, _, _, m in rec.items if k == "gemm" and Recorder.kernel_family(k, m) == 8)
        return f8, sum(fl for k, fl, _, _, m in rec.items if k in ("gemm", "flash_attn"))
    (f8a, falla), (f8b, fallb) = recorded(steps), recorded(2 * steps)
    f8, fall = f8b - f8a, fallb - falla
    print(json.dumps({"workload": f"SDXL-Turbo + Canny ControlNet, batch={b} {res}x{res}, {steps} DDIM steps, no CFG, ctrl-scale 0.75",
                      "fp8_flop_share_of_an_evaluation": round(f8 / fall, 4), "tflop_per_evaluation_per_image": round(fall / steps / b / 1e12, 3),
                      "images_per_s": round(b / dt, 3), "s_per_batch": round(dt, 3), "vae_decode_s_per_batch": round(tv, 3),
                      "vae_dtype": str(pipe.vae.dtype).replace("torch.", ""), "vae_gemm": pipe.vae.f32_gemm, "deterministic": bool(torch.equal(out, out2)),
                      "finite": bool(out.float().isfinite().all()), "dtype": "bf16 + fp8 (e4m3 W8A8) transformer projections" if "--fp8" in sys.argv else "bf16", "data": "synthetic"}), flush=True)
