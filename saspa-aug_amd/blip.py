"""BLIP-Diffusion subject front-end (a8) on the gfx950 kernels: the `Blip2QFormerModel` the
reference's config-3 pipeline runs once per variant to turn (reference image, subject category) into 16 subject
tokens for the context CLIP text encoder ([upstream] diffusers pipelines/blip_diffusion/modeling_blip2.py +
pipeline_blip_diffusion_controlnet.get_query_embeddings, recalled; reference call site run_aug/run_aug.py:243-250).

Launch graph only -- every matmul / norm / attention / activation is a C-ABI kernel of libsaspa_hip.so; the torch
calls below are views, concatenations and expands (data movement).  Weight key names are the checkpoint's."""
import numpy as np
import torch
from PIL import Image

from . import ops
from . import weights as W
from .models import _Packed, _f32, attention_core, project_vt


def preprocess_reference(img, cfg, mean, std):
    """BlipImageProcessor.preprocess: RGB, resize to image_size^2 (PIL bicubic), /255, normalise -> [3,S,S] fp32 (host)."""
    if not isinstance(img, Image.Image):
        img = Image.fromarray(np.asarray(img, dtype=np.uint8))
    s = cfg["image_size"]
    im = img.convert("RGB").resize((s, s), resample=Image.BICUBIC)
    x = torch.from_numpy(np.asarray(im, dtype=np.float32) / 255.0)
    x = (x - torch.tensor(mean)) / torch.tensor(std)
    return x.permute(2, 0, 1).contiguous()


class Blip2QFormer:
    def __init__(self, sd, cfg, dev, dtype):
        self.cfg, self.dev, self.dtype = cfg, dev, dtype
        pk = self.pk = _Packed(sd, dev, dtype)
        p = self.p = pk.p
        v = "visual_encoder"
        vw, w = cfg["vis_width"], cfg["width"]
        # patch embedding as a linear over unfolded 3*P*P patches (K padded to a multiple of 8)
        pw = sd[v + ".embeddings.patch_embedding.weight"].reshape(vw, -1)
        p["patch.w"] = W.pack_linear(pw).to(dev, dtype)
        pos = sd[v + ".embeddings.position_embedding"][0]
        p["cls_pos"] = (sd[v + ".embeddings.class_embedding"][0, 0] + pos[0]).to(dev, dtype).reshape(1, 1, vw).contiguous()
        p["patch_pos"] = pos[1:].to(dev, dtype).contiguous()
        pk.norm(v + ".pre_layernorm")
        for i in range(cfg["vis_layers"]):
            lp = f"{v}.encoder.layers.{i}"
            a = lp + ".self_attn"
            wqkv, bqkv = sd[a + ".qkv.weight"], sd[a + ".qkv.bias"]
            p[a + ".qk.w"] = wqkv[:2 * vw].contiguous().to(dev, dtype)
            p[a + ".qk.b"] = _f32(bqkv[:2 * vw], dev)
            p[a + ".v.w"] = wqkv[2 * vw:].contiguous().to(dev, dtype)
            wo = sd[a + ".projection.weight"]
            p[a + ".o.w"] = wo.contiguous().to(dev, dtype)
            p[a + ".o.b"] = _f32(sd[a + ".projection.bias"] + wo @ bqkv[2 * vw:], dev)   # softmax rows sum to 1
            pk.norm(lp + ".layer_norm1")
            pk.norm(lp + ".layer_norm2")
            pk.linear(lp + ".mlp.fc1")
            pk.linear(lp + ".mlp.fc2")
        pk.norm(v + ".post_layernorm")
        p["query_tokens"] = sd["query_tokens"].to(dev, dtype).contiguous()
        p["word"] = sd["embeddings.word_embeddings.weight"].to(dev, dtype).contiguous()
        p["tpos"] = sd["embeddings.position_embeddings.weight"].to(dev, dtype).contiguous()
        pk.norm("embeddings.LayerNorm")

        def attn(pfx):
            q, k, vv = (pfx + ".attention.query", pfx + ".attention.key", pfx + ".attention.value")
            if sd[q + ".weight"].shape[1] == sd[k + ".weight"].shape[1] and "crossattention" not in pfx:
                p[pfx + ".qk.w"] = torch.cat([sd[q + ".weight"], sd[k + ".weight"]], 0).contiguous().to(dev, dtype)
                p[pfx + ".qk.b"] = _f32(torch.cat([sd[q + ".bias"], sd[k + ".bias"]]), dev)
            else:
                pk.linear(q)
                pk.linear(k)
            p[pfx + ".v.w"] = W.pack_linear(sd[vv + ".weight"]).to(dev, dtype)
            wo = sd[pfx + ".output.dense.weight"]
            p[pfx + ".o.w"] = wo.contiguous().to(dev, dtype)
            p[pfx + ".o.b"] = _f32(sd[pfx + ".output.dense.bias"] + wo @ sd[vv + ".bias"], dev)
            pk.norm(pfx + ".output.LayerNorm")

        for i in range(cfg["layers"]):
            lp = f"encoder.layer.{i}"
            attn(lp + ".attention")
            if i % cfg["cross_freq"] == 0:
                attn(lp + ".crossattention")
            for sfx in ("", "_query"):
                pk.linear(f"{lp}.intermediate{sfx}.dense")
                pk.linear(f"{lp}.output{sfx}.dense")
                pk.norm(f"{lp}.output{sfx}.LayerNorm")
        pk.linear("proj_layer.dense1")
        pk.linear("proj_layer.dense2")
        pk.norm("proj_layer.LayerNorm")
        pk.sd = None

    # ---- vision tower ------------------------------------------------------------------
    def vision(self, pixel_values):
        """[B,3,S,S] (normalised, any float dtype) -> last_hidden_state [B, 1 + (S/P)^2, vis_width]."""
        cfg, p = self.cfg, self.p
        b, _, s, _ = pixel_values.shape
        ps, vw = cfg["patch"], cfg["vis_width"]
        g = s // ps
        x = ops.h2d(pixel_values, self.dev, self.dtype)
        patches = x.reshape(b, 3, g, ps, g, ps).permute(0, 2, 4, 1, 3, 5).reshape(b * g * g, 3 * ps * ps)
        kpad = p["patch.w"].shape[1] - patches.shape[1]
        if kpad:
            patches = torch.nn.functional.pad(patches, (0, kpad))
        pos = p["patch_pos"][None].expand(b, -1, -1).reshape(b * g * g, vw).contiguous()
        tok = ops.linear(patches.contiguous(), p["patch.w"], None, residual=pos).view(b, g * g, vw)
        x = torch.cat([p["cls_pos"].expand(b, -1, -1), tok], 1).contiguous()
        n = x.shape[1]
        v = "visual_encoder"
        eps = cfg["vis_eps"]
        x = ops.layernorm(x, p[v + ".pre_layernorm.g"], p[v + ".pre_layernorm.b"], eps)
        for i in range(cfg["vis_layers"]):
            lp = f"{v}.encoder.layers.{i}"
            a = lp + ".self_attn"
            h = ops.layernorm(x, p[lp + ".layer_norm1.g"], p[lp + ".layer_norm1.b"], eps)
            qk = ops.linear(h, p[a + ".qk.w"], p[a + ".qk.b"])
            vt = project_vt(h, p[a + ".v.w"], n)
            o = attention_core(qk[:, :, :vw], qk[:, :, vw:], vt, cfg["vis_heads"], n, n)
            x = ops.linear(o, p[a + ".o.w"], p[a + ".o.b"], residual=x)
            h = ops.layernorm(x, p[lp + ".layer_norm2.g"], p[lp + ".layer_norm2.b"], eps)
            h = ops.activation(ops.linear(h, p[lp + ".mlp.fc1.w"], p[lp + ".mlp.fc1.b"]), ops.ACT_QUICK_GELU)
            x = ops.linear(h, p[lp + ".mlp.fc2.w"], p[lp + ".mlp.fc2.b"], residual=x)
        return ops.layernorm(x, p[v + ".post_layernorm.g"], p[v + ".post_layernorm.b"], eps)

    # ---- Q-Former ----------------------------------------------------------------------
    def _attention(self, pfx, x, kv, heads):
        """BertSelfAttention + BertSelfOutput: LN(dense(attn(x, kv)) + x); kv is x itself or the image tokens."""
        p, w = self.p, self.cfg["width"]
        nq, nk = x.shape[1], kv.shape[1]
        if pfx + ".qk.w" in p:
            qk = ops.linear(x, p[pfx + ".qk.w"], p[pfx + ".qk.b"])
            q, k = qk[:, :, :w], qk[:, :, w:]
        else:
            q = ops.linear(x, p[pfx + ".attention.query.w"], p[pfx + ".attention.query.b"])
            k = ops.linear(kv, p[pfx + ".attention.key.w"], p[pfx + ".attention.key.b"])
        vt = project_vt(kv, p[pfx + ".v.w"], nk)
        o = attention_core(q, k, vt, heads, nq, nk)
        h = ops.linear(o, p[pfx + ".o.w"], p[pfx + ".o.b"], residual=x)
        return ops.layernorm(h, p[pfx + ".output.LayerNorm.g"], p[pfx + ".output.LayerNorm.b"], self.cfg["eps"])

    def _ffn(self, lp, sfx, x):
        p = self.p
        h = ops.activation(ops.linear(x, p[f"{lp}.intermediate{sfx}.dense.w"], p[f"{lp}.intermediate{sfx}.dense.b"]), ops.ACT_GELU)
        h = ops.linear(h, p[f"{lp}.output{sfx}.dense.w"], p[f"{lp}.output{sfx}.dense.b"], residual=x)
        return ops.layernorm(h, p[f"{lp}.output{sfx}.LayerNorm.g"], p[f"{lp}.output{sfx}.LayerNorm.b"], self.cfg["eps"])

    @torch.no_grad()
    def forward(self, pixel_values, input_ids):
        """pixel_values [B,3,S,S], input_ids int [B,T] ([CLS] category [SEP], same T for the batch) -> [B,nq,out_dim]."""
        cfg, p = self.cfg, self.p
        b, t = input_ids.shape
        nq, w = cfg["num_query"], cfg["width"]
        image_embeds = self.vision(pixel_values)
        ids = torch.as_tensor(np.asarray(input_ids)) if not torch.is_tensor(input_ids) else input_ids
        txt = ops.embed_tokens(ops.h2d(ids, self.dev), p["word"], p["tpos"], t).view(b, t, w)
        x = torch.cat([p["query_tokens"].expand(b, -1, -1), txt], 1).contiguous()
        x = ops.layernorm(x, p["embeddings.LayerNorm.g"], p["embeddings.LayerNorm.b"], cfg["eps"])
        for i in range(cfg["layers"]):
            lp = f"encoder.layer.{i}"
            x = self._attention(lp + ".attention", x, x, cfg["heads"])
            q, tx = x[:, :nq].contiguous(), x[:, nq:].contiguous()
            if i % cfg["cross_freq"] == 0:
                q = self._attention(lp + ".crossattention", q, image_embeds, cfg["heads"])
            q = self._ffn(lp, "_query", q)
            tx = self._ffn(lp, "", tx)
            x = torch.cat([q, tx], 1).contiguous()
        x_in = x[:, :nq].contiguous()
        h = ops.layernorm(x_in, p["proj_layer.LayerNorm.g"], p["proj_layer.LayerNorm.b"], cfg["eps"])
        h = ops.activation(ops.linear(h, p["proj_layer.dense1.w"], p["proj_layer.dense1.b"]), ops.ACT_QUICK_GELU)
        return ops.linear(h, p["proj_layer.dense2.w"], p["proj_layer.dense2.b"], residual=x_in)
