"""TEST INFRASTRUCTURE ONLY -- torch-CPU fp32 restatement of the diffusers
modules the reference's hot path executes (PARITY UNPINNED, see oracle/__init__).

Every function takes a flat ``state_dict`` whose keys are the diffusers
checkpoint names (so real ``safetensors`` files load unchanged) plus a key
prefix.  Layout is NCHW, dtype fp32, exactly like the diffusers CPU pipeline the
north-star names as the parity target.

Semantics followed ([upstream] diffusers 0.32.2, recalled; call site
run_aug/run_aug.py:278, construction :185/:206/:221):

* ``UNet2DConditionModel`` (SD-1.5 config)          -> :func:`unet_forward`
* ``ControlNetModel`` (control_v11p_sd15_canny)     -> :func:`controlnet_forward`
* ``AutoencoderKL.decode``                          -> :func:`vae_decode`
* ``CLIPTextModel`` (ViT-L/14 text tower)           -> :func:`clip_text_forward`
"""
import math

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# configs (SD-v1.5 family).  "heads" is diffusers' mis-named
# ``attention_head_dim=8`` (= number of heads; head dim = C / 8).
# --------------------------------------------------------------------------
SD15_UNET = dict(
    in_channels=4, out_channels=4, block_out=(320, 640, 1280, 1280),
    attn=(True, True, True, False), layers=2, heads=8, ctx_dim=768, groups=32,
    temb_dim=1280,
)
SD15_CONTROLNET = dict(SD15_UNET, cond_channels=3, cond_embed=(16, 32, 96, 256))
SD15_VAE = dict(latent_channels=4, out_channels=3, block_out=(128, 256, 512, 512),
                layers=2, groups=32, scaling_factor=0.18215)
CLIP_L = dict(vocab=49408, width=768, layers=12, heads=12, mlp=3072, max_pos=77)


def _p(sd, name):
    return sd[name]


def conv2d(sd, pfx, x, stride=1, padding=1):
    return F.conv2d(x, _p(sd, pfx + ".weight"), sd.get(pfx + ".bias"), stride=stride, padding=padding)


def linear(sd, pfx, x):
    return F.linear(x, _p(sd, pfx + ".weight"), sd.get(pfx + ".bias"))


def group_norm(sd, pfx, x, groups, eps):
    return F.group_norm(x, groups, _p(sd, pfx + ".weight"), _p(sd, pfx + ".bias"), eps)


def layer_norm(sd, pfx, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), _p(sd, pfx + ".weight"), _p(sd, pfx + ".bias"), eps)


# --------------------------------------------------------------------------
# timestep embedding: Timesteps(320, flip_sin_to_cos=True, freq_shift=0)
# --------------------------------------------------------------------------
def timestep_sinusoid(t, dim, max_period=10000.0):
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32) / half
    emb = t.float()[:, None] * torch.exp(exponent)[None, :]
    # diffusers builds cat([sin, cos]) then flips halves when flip_sin_to_cos
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def time_embedding(sd, pfx, t, dim0):
    e = timestep_sinusoid(t, dim0)
    e = linear(sd, pfx + ".linear_1", e)
    e = F.silu(e)
    return linear(sd, pfx + ".linear_2", e)


# --------------------------------------------------------------------------
# ResnetBlock2D
# --------------------------------------------------------------------------
def resnet_block(sd, pfx, x, temb, groups, eps):
    h = group_norm(sd, pfx + ".norm1", x, groups, eps)
    h = F.silu(h)
    h = conv2d(sd, pfx + ".conv1", h)
    if temb is not None and (pfx + ".time_emb_proj.weight") in sd:
        h = h + linear(sd, pfx + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = group_norm(sd, pfx + ".norm2", h, groups, eps)
    h = F.silu(h)
    h = conv2d(sd, pfx + ".conv2", h)
    if (pfx + ".conv_shortcut.weight") in sd:
        x = conv2d(sd, pfx + ".conv_shortcut", x, padding=0)
    return x + h


# --------------------------------------------------------------------------
# attention (AttnProcessor2_0 == softmax(q k^T / sqrt(d)) v, no upcast)
# --------------------------------------------------------------------------
def mha(q, k, v, heads, mask=None):
    b, n, c = q.shape
    d = c // heads
    q = q.view(b, n, heads, d).transpose(1, 2)
    k = k.view(b, k.shape[1], heads, d).transpose(1, 2)
    v = v.view(b, v.shape[1], heads, d).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
    if mask is not None:
        s = s + mask
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, v)
    return o.transpose(1, 2).reshape(b, n, c)


def attention(sd, pfx, x, ctx, heads):
    ctx = x if ctx is None else ctx
    q = linear(sd, pfx + ".to_q", x)
    k = linear(sd, pfx + ".to_k", ctx)
    v = linear(sd, pfx + ".to_v", ctx)
    o = mha(q, k, v, heads)
    return linear(sd, pfx + ".to_out.0", o)


def basic_transformer_block(sd, pfx, h, ctx, heads):
    n = layer_norm(sd, pfx + ".norm1", h)
    h = attention(sd, pfx + ".attn1", n, None, heads) + h
    n = layer_norm(sd, pfx + ".norm2", h)
    h = attention(sd, pfx + ".attn2", n, ctx, heads) + h
    n = layer_norm(sd, pfx + ".norm3", h)
    g = linear(sd, pfx + ".ff.net.0.proj", n)
    val, gate = g.chunk(2, dim=-1)
    g = val * F.gelu(gate)            # erf GELU (GEGLU)
    h = linear(sd, pfx + ".ff.net.2", g) + h
    return h


def transformer_2d(sd, pfx, x, ctx, heads, groups, depth=1, linear_proj=False):
    """Transformer2DModel: conv projections around ONE block (SD-1.5, use_linear_projection=False) or linear
    projections applied to the token matrix around ``depth`` blocks (SDXL, use_linear_projection=True)."""
    b, c, hh, ww = x.shape
    res = x
    h = group_norm(sd, pfx + ".norm", x, groups, 1e-6)
    if linear_proj:
        h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
        h = linear(sd, pfx + ".proj_in", h)
    else:
        h = conv2d(sd, pfx + ".proj_in", h, padding=0)
        h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    for d in range(depth):
        h = basic_transformer_block(sd, f"{pfx}.transformer_blocks.{d}", h, ctx, heads)
    if linear_proj:
        h = linear(sd, pfx + ".proj_out", h)
        h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
    else:
        h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
        h = conv2d(sd, pfx + ".proj_out", h, padding=0)
    return h + res


def _lvl(cfg, i):
    """(heads, depth) of level i: SD-1.5 has one head count and depth 1; SDXL has per-level tuples."""
    heads = cfg["heads"][i] if isinstance(cfg["heads"], (tuple, list)) else cfg["heads"]
    depth = cfg["depth"][i] if "depth" in cfg else 1
    return heads, depth


def _tr(sd, cfg, pfx, x, ctx, lvl):
    heads, depth = _lvl(cfg, lvl)
    return transformer_2d(sd, pfx, x, ctx, heads, cfg["groups"], depth, cfg.get("linear_proj", False))


def added_embedding(sd, cfg, text_embeds, time_ids):
    """SDXL addition_embed_type "text_time" (UNet2DConditionModel.get_aug_embed): each of the 6 size / crop ids
    through Timesteps(256, flip_sin_to_cos=True, freq_shift=0), flattened, appended AFTER the pooled text
    embedding, then TimestepEmbedding(2816 -> 1280).  text_embeds [B, pooled], time_ids [B, 6]."""
    ae = cfg["add_embed"]
    b = text_embeds.shape[0]
    te = timestep_sinusoid(time_ids.reshape(-1), ae["time_dim"]).reshape(b, -1)
    e = torch.cat([text_embeds, te], dim=-1)
    e = F.silu(linear(sd, "add_embedding.linear_1", e))
    return linear(sd, "add_embedding.linear_2", e)


def _emb(sd, cfg, t, batch, added):
    t = torch.as_tensor(t).reshape(-1).expand(batch)
    emb = time_embedding(sd, "time_embedding", t, cfg["block_out"][0])
    if "add_embed" in cfg:
        emb = emb + added_embedding(sd, cfg, added["text_embeds"], added["time_ids"])
    return emb


# --------------------------------------------------------------------------
# encoder shared by UNet and ControlNet
# --------------------------------------------------------------------------
def _encoder(sd, cfg, sample, emb, ctx):
    res = (sample,)
    n_lvl = len(cfg["block_out"])
    for i in range(n_lvl):
        for j in range(cfg["layers"]):
            sample = resnet_block(sd, f"down_blocks.{i}.resnets.{j}", sample, emb, cfg["groups"], 1e-5)
            if cfg["attn"][i]:
                sample = _tr(sd, cfg, f"down_blocks.{i}.attentions.{j}", sample, ctx, i)
            res += (sample,)
        if i != n_lvl - 1:
            sample = conv2d(sd, f"down_blocks.{i}.downsamplers.0.conv", sample, stride=2, padding=1)
            res += (sample,)
    sample = resnet_block(sd, "mid_block.resnets.0", sample, emb, cfg["groups"], 1e-5)
    sample = _tr(sd, cfg, "mid_block.attentions.0", sample, ctx, n_lvl - 1)
    sample = resnet_block(sd, "mid_block.resnets.1", sample, emb, cfg["groups"], 1e-5)
    return sample, res


def controlnet_cond_embedding(sd, cfg, cond):
    pfx = "controlnet_cond_embedding"
    e = F.silu(conv2d(sd, pfx + ".conv_in", cond))
    n = len(cfg["cond_embed"])
    for i in range(n - 1):
        e = F.silu(conv2d(sd, f"{pfx}.blocks.{2 * i}", e))
        e = F.silu(conv2d(sd, f"{pfx}.blocks.{2 * i + 1}", e, stride=2))
    return conv2d(sd, pfx + ".conv_out", e)


def controlnet_forward(sd, cfg, sample, t, ctx, cond, conditioning_scale=1.0, added=None):
    """ControlNetModel.forward: returns (12 down residuals, mid residual), each
    already multiplied by ``conditioning_scale`` (run_aug/run_aug.py:269 passes 0.75).
    ``added`` = dict(text_embeds, time_ids) for the SDXL text_time conditioning."""
    emb = _emb(sd, cfg, t, sample.shape[0], added)
    sample = conv2d(sd, "conv_in", sample)
    sample = sample + controlnet_cond_embedding(sd, cfg, cond)
    mid, res = _encoder(sd, cfg, sample, emb, ctx)
    out = []
    for i, r in enumerate(res):
        out.append(conv2d(sd, f"controlnet_down_blocks.{i}", r, padding=0) * conditioning_scale)
    mid = conv2d(sd, "controlnet_mid_block", mid, padding=0) * conditioning_scale
    return out, mid


def unet_forward(sd, cfg, sample, t, ctx, down_residuals=None, mid_residual=None, added=None):
    """UNet2DConditionModel.forward with ControlNet residuals added to the 12
    skips and to the mid-block output."""
    emb = _emb(sd, cfg, t, sample.shape[0], added)
    sample = conv2d(sd, "conv_in", sample)
    sample, res = _encoder(sd, cfg, sample, emb, ctx)
    if down_residuals is not None:
        res = tuple(r + a for r, a in zip(res, down_residuals))
    if mid_residual is not None:
        sample = sample + mid_residual
    n_lvl = len(cfg["block_out"])
    rev_attn = tuple(reversed(cfg["attn"]))
    for i in range(n_lvl):
        for j in range(cfg["layers"] + 1):
            skip = res[-1]
            res = res[:-1]
            sample = torch.cat([sample, skip], dim=1)
            sample = resnet_block(sd, f"up_blocks.{i}.resnets.{j}", sample, emb, cfg["groups"], 1e-5)
            if rev_attn[i]:
                sample = _tr(sd, cfg, f"up_blocks.{i}.attentions.{j}", sample, ctx, n_lvl - 1 - i)
        if i != n_lvl - 1:
            sample = F.interpolate(sample, scale_factor=2.0, mode="nearest")
            sample = conv2d(sd, f"up_blocks.{i}.upsamplers.0.conv", sample)
    sample = group_norm(sd, "conv_norm_out", sample, cfg["groups"], 1e-5)
    sample = F.silu(sample)
    return conv2d(sd, "conv_out", sample)


# --------------------------------------------------------------------------
# AutoencoderKL.decode
# --------------------------------------------------------------------------
def vae_mid_attention(sd, pfx, x, groups):
    b, c, hh, ww = x.shape
    res = x
    h = group_norm(sd, pfx + ".group_norm", x, groups, 1e-6)
    h = h.view(b, c, hh * ww).transpose(1, 2)
    q = linear(sd, pfx + ".to_q", h)
    k = linear(sd, pfx + ".to_k", h)
    v = linear(sd, pfx + ".to_v", h)
    o = mha(q, k, v, 1)
    o = linear(sd, pfx + ".to_out.0", o)
    o = o.transpose(1, 2).reshape(b, c, hh, ww)
    return o + res


def vae_decode(sd, cfg, z):
    """AutoencoderKL.decode(z).sample; caller divides latents by scaling_factor."""
    g = cfg["groups"]
    z = conv2d(sd, "post_quant_conv", z, padding=0)
    h = conv2d(sd, "decoder.conv_in", z)
    h = resnet_block(sd, "decoder.mid_block.resnets.0", h, None, g, 1e-6)
    h = vae_mid_attention(sd, "decoder.mid_block.attentions.0", h, g)
    h = resnet_block(sd, "decoder.mid_block.resnets.1", h, None, g, 1e-6)
    n_lvl = len(cfg["block_out"])
    for i in range(n_lvl):
        for j in range(cfg["layers"] + 1):
            h = resnet_block(sd, f"decoder.up_blocks.{i}.resnets.{j}", h, None, g, 1e-6)
        if i != n_lvl - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = conv2d(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", h)
    h = group_norm(sd, "decoder.conv_norm_out", h, g, 1e-6)
    h = F.silu(h)
    return conv2d(sd, "decoder.conv_out", h)


def vae_encode(sd, cfg, x):
    """AutoencoderKL.encode(x).latent_dist parameters: x [B,3,H,W] in [-1,1] -> (mean, logvar) [B,4,H/8,W/8] each
    (Encoder: conv_in, 4 down blocks of 2 resnets + Downsample2D(padding=0: F.pad (0,1,0,1) then stride-2 conv) on the first
    three, mid block with one single-head attention, GroupNorm + SiLU, conv_out to 2*latent channels; quant_conv)."""
    g = cfg["groups"]
    h = conv2d(sd, "encoder.conv_in", x)
    n_lvl = len(cfg["block_out"])
    for i in range(n_lvl):
        for j in range(cfg["layers"]):
            h = resnet_block(sd, f"encoder.down_blocks.{i}.resnets.{j}", h, None, g, 1e-6)
        if i != n_lvl - 1:
            h = F.pad(h, (0, 1, 0, 1))
            h = conv2d(sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", h, stride=2, padding=0)
    h = resnet_block(sd, "encoder.mid_block.resnets.0", h, None, g, 1e-6)
    h = vae_mid_attention(sd, "encoder.mid_block.attentions.0", h, g)
    h = resnet_block(sd, "encoder.mid_block.resnets.1", h, None, g, 1e-6)
    h = F.silu(group_norm(sd, "encoder.conv_norm_out", h, g, 1e-6))
    h = conv2d(sd, "encoder.conv_out", h)
    moments = conv2d(sd, "quant_conv", h, padding=0)
    return moments.chunk(2, dim=1)


# --------------------------------------------------------------------------
# CLIP text tower (last_hidden_state after final LN; causal mask; quick-GELU)
# --------------------------------------------------------------------------
def clip_text_forward(sd, cfg, ids, ctx_embeddings=None, ctx_begin_pos=2, penultimate=False):
    """CLIPTextModel; with ``ctx_embeddings`` [B, nctx, width] it is BLIP-Diffusion's ContextCLIPTextModel
    ([upstream] diffusers blip_diffusion/modeling_ctx_clip.py, recalled): the subject tokens are spliced into the
    token embeddings at ``ctx_begin_pos`` BEFORE the position embeddings (of the lengthened sequence) are added."""
    pfx = "text_model"
    tok = sd[pfx + ".embeddings.token_embedding.weight"][ids]
    if ctx_embeddings is not None:
        tok = torch.cat([tok[:, :ctx_begin_pos], ctx_embeddings.to(tok.dtype), tok[:, ctx_begin_pos:]], dim=1)
    b, n = tok.shape[:2]
    x = tok + sd[pfx + ".embeddings.position_embedding.weight"][:n][None]
    mask = torch.full((n, n), float("-inf")).triu_(1)
    hidden = [x]
    for i in range(cfg["layers"]):
        lp = f"{pfx}.encoder.layers.{i}"
        h = layer_norm(sd, lp + ".layer_norm1", x)
        q = linear(sd, lp + ".self_attn.q_proj", h)
        k = linear(sd, lp + ".self_attn.k_proj", h)
        v = linear(sd, lp + ".self_attn.v_proj", h)
        o = mha(q, k, v, cfg["heads"], mask)
        x = x + linear(sd, lp + ".self_attn.out_proj", o)
        h = layer_norm(sd, lp + ".layer_norm2", x)
        h = linear(sd, lp + ".mlp.fc1", h)
        h = F.gelu(h) if cfg.get("act") == "gelu" else h * torch.sigmoid(1.702 * h)
        x = x + linear(sd, lp + ".mlp.fc2", h)
        hidden.append(x)
    last = layer_norm(sd, pfx + ".final_layer_norm", x)
    if not penultimate:
        return last
    # SDXL encode_prompt: hidden_states[-2] of the tower (no final LayerNorm); for CLIPTextModelWithProjection
    # also text_embeds = text_projection(last_hidden_state[eos]), eos = first position holding the largest id
    # (eos_token_id == 2 legacy branch of transformers' CLIPTextTransformer: input_ids.argmax(-1)).
    pooled = None
    if "text_projection.weight" in sd:
        eos = ids.argmax(dim=-1)
        pooled = F.linear(last[torch.arange(b), eos], sd["text_projection.weight"])
    return hidden[-2], pooled
