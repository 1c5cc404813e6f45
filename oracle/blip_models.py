"""TEST INFRASTRUCTURE ONLY -- torch-CPU fp32 restatement of the BLIP-Diffusion subject front-end the
reference's config-3 path executes once per variant (run_aug/run_aug.py:243-250 -> [upstream]
``BlipDiffusionControlNetPipeline.get_query_embeddings`` -> ``Blip2QFormerModel``; diffusers 0.32.2
pipelines/blip_diffusion/modeling_blip2.py, recalled).  PARITY UNPINNED (see oracle/__init__).

``blip2_qformer_forward(sd, cfg, pixel_values, input_ids)``: normalised 224x224 reference image + BERT token ids of
the subject category -> [B, 16, 768] subject tokens for ContextCLIPTextModel (oracle/sd_models.clip_text_forward)."""
import torch
import torch.nn.functional as F

from .sd_models import layer_norm, linear, mha


def preprocess_reference(img_u8_hwc, cfg, mean, std):
    """BlipImageProcessor: resize to image_size x image_size (bicubic, PIL), /255, normalise; -> [1,3,S,S] fp32."""
    from PIL import Image
    import numpy as np
    s = cfg["image_size"]
    im = Image.fromarray(img_u8_hwc).convert("RGB").resize((s, s), resample=Image.BICUBIC)
    x = torch.from_numpy(np.asarray(im, dtype=np.float32) / 255.0)
    x = (x - torch.tensor(mean)) / torch.tensor(std)
    return x.permute(2, 0, 1)[None].contiguous()


def blip2_vision_forward(sd, cfg, pixel_values):
    """Blip2VisionModel.last_hidden_state (pre-LN ViT, quick-GELU MLP, fused qkv; post_layernorm on every token)."""
    v = "visual_encoder"
    b = pixel_values.shape[0]
    x = F.conv2d(pixel_values, sd[v + ".embeddings.patch_embedding.weight"], None, stride=cfg["patch"])
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([sd[v + ".embeddings.class_embedding"].expand(b, -1, -1), x], 1)
    x = x + sd[v + ".embeddings.position_embedding"][:, :x.shape[1]]
    x = layer_norm(sd, v + ".pre_layernorm", x, cfg["vis_eps"])
    w = cfg["vis_width"]
    for i in range(cfg["vis_layers"]):
        lp = f"{v}.encoder.layers.{i}"
        h = layer_norm(sd, lp + ".layer_norm1", x, cfg["vis_eps"])
        qkv = linear(sd, lp + ".self_attn.qkv", h)
        o = mha(qkv[..., :w], qkv[..., w:2 * w], qkv[..., 2 * w:], cfg["vis_heads"])
        x = x + linear(sd, lp + ".self_attn.projection", o)
        h = layer_norm(sd, lp + ".layer_norm2", x, cfg["vis_eps"])
        h = linear(sd, lp + ".mlp.fc1", h)
        h = h * torch.sigmoid(1.702 * h)
        x = x + linear(sd, lp + ".mlp.fc2", h)
    return layer_norm(sd, v + ".post_layernorm", x, cfg["vis_eps"])


def _bert_attention(sd, pfx, x, kv, heads, eps):
    """BertSelfAttention + BertSelfOutput (post-LN): LN(dense(attn(x, kv)) + x)."""
    q = linear(sd, pfx + ".attention.query", x)
    k = linear(sd, pfx + ".attention.key", kv)
    v = linear(sd, pfx + ".attention.value", kv)
    o = mha(q, k, v, heads)
    return layer_norm(sd, pfx + ".output.LayerNorm", linear(sd, pfx + ".output.dense", o) + x, eps)


def _bert_ffn(sd, lp, sfx, x, eps):
    h = F.gelu(linear(sd, f"{lp}.intermediate{sfx}.dense", x))
    return layer_norm(sd, f"{lp}.output{sfx}.LayerNorm", linear(sd, f"{lp}.output{sfx}.dense", h) + x, eps)


def qformer_encoder(sd, cfg, x, image_embeds):
    """Blip2QFormerModel's embedding LayerNorm + encoder on the concatenated [queries | text] embeddings `x` [B, nq + T, W]
    with the image tokens as cross-attention memory: self-attention over all tokens, cross-attention for the query part
    every `cross_freq` layers, separate feed-forward weights for the query part (`*_query`) and the text part.
    PINNED against transformers.Blip2QFormerModel on the same state dict (tests/test_oracle.py)."""
    nq, eps = cfg["num_query"], cfg["eps"]
    x = layer_norm(sd, "embeddings.LayerNorm", x, eps)
    for i in range(cfg["layers"]):
        lp = f"encoder.layer.{i}"
        x = _bert_attention(sd, lp + ".attention", x, x, cfg["heads"], eps)          # queries and text attend jointly
        q, tx = x[:, :nq], x[:, nq:]
        if i % cfg["cross_freq"] == 0:
            q = _bert_attention(sd, lp + ".crossattention", q, image_embeds, cfg["heads"], eps)
        q = _bert_ffn(sd, lp, "_query", q, eps)
        tx = _bert_ffn(sd, lp, "", tx, eps)
        x = torch.cat([q, tx], 1)
    return x


def blip2_qformer_forward(sd, cfg, pixel_values, input_ids):
    """Blip2QFormerModel.forward(image_input, text_input, return_dict=False): proj_layer(sequence_output[:, :nq])."""
    b, t = input_ids.shape
    nq, eps = cfg["num_query"], cfg["eps"]
    image_embeds = blip2_vision_forward(sd, cfg, pixel_values)
    txt = sd["embeddings.word_embeddings.weight"][input_ids] + sd["embeddings.position_embeddings.weight"][:t][None]
    x = torch.cat([sd["query_tokens"].expand(b, -1, -1), txt], 1)
    x = qformer_encoder(sd, cfg, x, image_embeds)
    x_in = x[:, :nq]
    h = layer_norm(sd, "proj_layer.LayerNorm", x_in, eps)
    h = linear(sd, "proj_layer.dense1", h)
    h = h * torch.sigmoid(1.702 * h)
    return linear(sd, "proj_layer.dense2", h) + x_in
