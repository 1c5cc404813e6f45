"""Architecture configs of the model families the reference's hot path instantiates
(run_aug/run_aug.py:53-72 BASE_MODEL_DICT / CONTROLNET_DICT_SD -> HF repo ids; the numbers
below are those repos' public config.json values)."""

# runwayml/stable-diffusion-v1-5 : unet/config.json.  "heads" is diffusers'
# attention_head_dim=8, which for this model is the NUMBER of heads (head dim = C/8).
SD15_UNET = dict(
    in_channels=4, out_channels=4, block_out=(320, 640, 1280, 1280),
    attn=(True, True, True, False), layers=2, heads=8, ctx_dim=768, groups=32, temb_dim=1280,
)
# lllyasviel/control_v11p_sd15_canny : config.json
SD15_CONTROLNET = dict(SD15_UNET, cond_channels=3, cond_embed=(16, 32, 96, 256))
# runwayml/stable-diffusion-v1-5 : vae/config.json (decoder half)
SD15_VAE = dict(latent_channels=4, out_channels=3, block_out=(128, 256, 512, 512), layers=2, groups=32,
                scaling_factor=0.18215)
# openai/clip-vit-large-patch14 text tower (text_encoder/config.json)
CLIP_L = dict(vocab=49408, width=768, layers=12, heads=12, mlp=3072, max_pos=77)

# runwayml/stable-diffusion-v1-5 : safety_checker/config.json (StableDiffusionSafetyChecker = CLIP ViT-L/14 vision tower
# + visual_projection 1024 -> 768 + 17 concept / 3 special-care embeddings with thresholds) and feature_extractor
# (CLIPImageProcessor: shortest edge 224 bicubic, centre crop 224, CLIP mean / std).  The reference never disables it
# (run_aug/run_aug.py:185-207).
SAFETY_CHECKER = dict(image_size=224, patch=14, width=1024, layers=24, heads=16, mlp=4096, proj_dim=768, n_concepts=17,
                      n_special=3)
SD15 = dict(unet=SD15_UNET, controlnet=SD15_CONTROLNET, vae=SD15_VAE, text=CLIP_L, safety=SAFETY_CHECKER)

# stabilityai/sdxl-turbo (= SDXL-base architecture) : unet/config.json.  Three levels, no attention at level 0,
# transformer_layers_per_block (1, 2, 10) -> "depth" (level 0 unused), attention_head_dim (5, 10, 20) = heads per
# level (head dim 64 everywhere), cross_attention_dim 2048 (CLIP-L 768 | OpenCLIP-bigG 1280), use_linear_projection,
# addition_embed_type "text_time": 6 size/crop ids x Timesteps(256) | pooled bigG text embedding (1280) -> 2816 -> 1280.
# Values recalled from the public repos (no network here): parity unpinned.
SDXL_UNET = dict(
    in_channels=4, out_channels=4, block_out=(320, 640, 1280), attn=(False, True, True), depth=(0, 2, 10),
    layers=2, heads=(5, 10, 20), ctx_dim=2048, groups=32, temb_dim=1280, linear_proj=True,
    add_embed=dict(time_dim=256, n_ids=6, pooled_dim=1280),
)
# diffusers/controlnet-canny-sdxl-1.0 : config.json (run_aug/run_aug.py:70)
SDXL_CONTROLNET = dict(SDXL_UNET, cond_channels=3, cond_embed=(16, 32, 96, 256))
# madebyollin/sdxl-vae-fp16-fix (run_aug/run_aug.py:189): the SD VAE architecture, scaling_factor 0.13025
SDXL_VAE = dict(SD15_VAE, scaling_factor=0.13025)
# text_encoder: CLIP-L as above; text_encoder_2: CLIPTextModelWithProjection (OpenCLIP ViT-bigG/14 text tower):
# 32 layers, width 1280, 20 heads, MLP 5120, erf-GELU, pad token id 0, text_projection 1280 -> 1280 (no bias).
# The SDXL pipelines read hidden_states[-2] of BOTH towers (output of the second-to-last layer, no final LayerNorm).
OPENCLIP_BIGG = dict(vocab=49408, width=1280, layers=32, heads=20, mlp=5120, max_pos=77, act="gelu", proj_dim=1280, pad_id=0)
SDXL_TURBO = dict(unet=SDXL_UNET, controlnet=SDXL_CONTROLNET, vae=SDXL_VAE, text=CLIP_L, text2=OPENCLIP_BIGG)

# Salesforce/blipdiffusion(-controlnet) : qformer/config.json (Blip2QFormerModel = CLIP-L/14 vision tower with its
# last block dropped + BERT-base Q-Former with 16 learned queries + ProjLayer), image processor mean/std, ctx_begin_pos.
# Values recalled from the public repo / LAVIS BlipDiffusion defaults (no network here): parity unpinned.
BLIP2_QFORMER = dict(
    image_size=224, patch=14, vis_width=1024, vis_layers=23, vis_heads=16, vis_mlp=4096, vis_eps=1e-5,
    width=768, layers=12, heads=12, mlp=3072, cross_freq=1, num_query=16, vocab=30523, max_pos=512, eps=1e-12,
    proj_hidden=3072, out_dim=768,
)
BLIP_IMAGE_MEAN = (0.48145466, 0.4578275, 0.40821073)
BLIP_IMAGE_STD = (0.26862954, 0.26130258, 0.27577711)
BLIP_DIFFUSION = dict({k: v for k, v in SD15.items() if k != "safety"}, qformer=BLIP2_QFORMER, ctx_begin_pos=2)


# ---- filter stage (all_utils/utils.py:252-255, :306-323, :357-375) ----
# OpenAI CLIP "RN50" (clip.load('RN50')): ModifiedResNet (3, 4, 6, 3) width 64 -> 2048 channels at 7x7, attention pool with
# 32 heads -> 1024-d embedding; text tower 12 layers x 512 wide x 8 heads, context 77.  Recalled from the public model.py.
CLIP_RN50 = dict(image_size=224, layers=(3, 4, 6, 3), width=64, heads=32, embed_dim=1024, vocab=49408, text_width=512,
                 text_heads=8, text_layers=12, context=77)
# the baseline classifier: WSDAN_CAL on ResNet-101 features (fgvc/models/cal.py:131-166; resnet50 is the fallback of
# BaseUtils.load_baseline_model, all_utils/dataset_utils.py:101-109), 32 attention maps, 224 x 224 crops
WSDAN_CAL_R101 = dict(image_size=224, layers=(3, 4, 23, 3), width=64, attentions=32)
WSDAN_CAL_R50 = dict(image_size=224, layers=(3, 4, 6, 3), width=64, attentions=32)


# controlnet_aux HEDdetector network (ControlNetHED_Apache2: VGG-16 conv stack, one 1x1 side output per block); the
# reference builds it for CONTROLNET = "hed" (run_aug/run_aug.py:311-312)
HED = dict(blocks=((64, 2), (128, 2), (256, 3), (512, 3), (512, 3)))
HED_TINY = dict(blocks=((8, 2), (16, 2), (16, 3), (24, 3), (24, 3)))


def tiny_filters(num_classes=12):
    """Reduced-width filter models with the same topology (tests)."""
    rn = dict(image_size=64, layers=(1, 1, 1, 1), width=16, heads=4, embed_dim=32, vocab=512, text_width=32, text_heads=2,
              text_layers=2, context=77)
    cal = dict(image_size=64, layers=(1, 1, 2, 1), width=16, attentions=8, num_classes=num_classes)
    return dict(clip_rn50=rn, cal=cal)


def tiny_xl(width=32, ctx1=32, ctx2=64, groups=8, vae_width=16):
    """Reduced-width SDXL family with the same topology: 3 levels, no attention at level 0, transformer depths
    (2, 3), linear projections, head dim 16, two text towers (ctx1 | ctx2), text_time addition embedding."""
    unet = dict(in_channels=4, out_channels=4, block_out=(width, 2 * width, 4 * width), attn=(False, True, True),
                depth=(0, 2, 3), layers=2, heads=(2, 4, 8), ctx_dim=ctx1 + ctx2, groups=groups, temb_dim=4 * width,
                linear_proj=True, add_embed=dict(time_dim=16, n_ids=6, pooled_dim=ctx2))
    cn = dict(unet, cond_channels=3, cond_embed=(8, 16, 24, 32))
    vae = dict(latent_channels=4, out_channels=3, block_out=(vae_width, 2 * vae_width, 4 * vae_width, 4 * vae_width),
               layers=2, groups=groups, scaling_factor=0.13025)
    text = dict(vocab=512, width=ctx1, layers=3, heads=2, mlp=4 * ctx1, max_pos=77)
    text2 = dict(vocab=512, width=ctx2, layers=3, heads=4, mlp=4 * ctx2, max_pos=77, act="gelu", proj_dim=ctx2, pad_id=0)
    return dict(unet=unet, controlnet=cn, vae=vae, text=text, text2=text2)


def tiny(width=32, ctx=64, groups=8, heads=4, vae_width=16):
    """Reduced-width family with the same topology (tests / smoke)."""
    unet = dict(in_channels=4, out_channels=4, block_out=(width, 2 * width, 4 * width, 4 * width),
                attn=(True, True, True, False), layers=2, heads=heads, ctx_dim=ctx, groups=groups,
                temb_dim=4 * width)
    cn = dict(unet, cond_channels=3, cond_embed=(8, 16, 24, 32))
    vae = dict(latent_channels=4, out_channels=3, block_out=(vae_width, 2 * vae_width, 4 * vae_width, 4 * vae_width),
               layers=2, groups=groups, scaling_factor=0.18215)
    text = dict(vocab=512, width=ctx, layers=2, heads=4, mlp=4 * ctx, max_pos=77)
    qformer = dict(image_size=56, patch=14, vis_width=64, vis_layers=2, vis_heads=4, vis_mlp=128, vis_eps=1e-5,
                   width=ctx, layers=2, heads=4, mlp=2 * ctx, cross_freq=1, num_query=16, vocab=512, max_pos=32, eps=1e-12,
                   proj_hidden=2 * ctx, out_dim=ctx)
    safety = dict(image_size=56, patch=14, width=64, layers=2, heads=4, mlp=128, proj_dim=48, n_concepts=17, n_special=3)
    return dict(unet=unet, controlnet=cn, vae=vae, text=text, qformer=qformer, ctx_begin_pos=2, safety=safety)
