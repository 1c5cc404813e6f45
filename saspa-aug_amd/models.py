"""The diffusers modules the reference's `pipe(...)` call executes (run_aug/run_aug.py:278),
re-expressed as launch sequences of the gfx950 kernels (saspa_aug_amd.ops).

Layout decisions (MI355X-first, not a translation of diffusers' NCHW modules):
  * activations are channels-last [B,H,W,C] so a transformer's token matrix [B,HW,C] is the
    same buffer as the conv feature map -- no permutes anywhere;
  * every conv / linear is one implicit-GEMM launch with bias, time-embedding row vector,
    residual add, SiLU and the ControlNet conditioning scale fused in its epilogue;
  * nearest-x2 upsampling and the skip concat are folded into the conv's A-operand loader;
  * q,k projections are one fused GEMM; the v projection is computed TRANSPOSED (swapped
    GEMM operands) so flash attention never transposes; v biases are folded into the
    out-projection bias (softmax rows sum to 1);
  * time-invariant work is hoisted out of the step loop: the time-embedding MLP and every
    resnet's time_emb_proj for ALL steps (one [steps, C] table each), cross-attention K/V
    per prompt, the ControlNet conditioning embedding per image;
  * ControlNet residuals are added to the UNet skips inside the ControlNet zero-conv
    epilogue (residual = UNet skip), so the UNet decoder reads the summed tensors.

Semantics follow [upstream] diffusers 0.32.2 as listed in SURVEY.md 3.2 / 8(a7)."""
import math
import os

import numpy as np
import torch

from . import ops
from . import weights as W

SILU = ops.ACT_SILU


def _f32(t, dev):
    return t.detach().to(dev, torch.float32).contiguous()


class _Packed:
    """Device-resident, kernel-layout parameters of one network."""

    def __init__(self, sd, dev, dtype):
        self.sd, self.dev, self.dtype = sd, dev, dtype
        self.p = {}

    def conv(self, name, split=None):
        w = self.sd[name + ".weight"]
        if w.dim() == 2:            # use_linear_projection (SDXL): a Linear over the token matrix == a 1x1 conv
            w = w[:, :, None, None]
        kh, kw = w.shape[2], w.shape[3]
        if split is None:
            pk = W.pack_conv(w)
            c0p, c1p = W.round8(w.shape[1]), 0
        else:
            c0, c1 = split
            c0p, c1p = W.round8(c0), W.round8(c1)
            pk = W.pack_conv_split(w, c0, c0p, c1, c1p)
        chunk = W.chunk_major_ok(kh, kw, c0p, c1p, self.dtype)
        if chunk:       # the taps of one channel chunk become consecutive K-tiles (input rows re-read from L2)
            pk = W.to_chunk_major(pk, kh * kw, self.dtype)
        t = pk.to(self.dev, self.dtype)
        t.saspa_korder = 1 if chunk else 0          # read by ops.conv -> SaspaGemmParams.korder
        t.saspa_cin = c0p + c1p                     # padded input channels per tap (VAEDecoder._presplit)
        self.p[name + ".w"] = t
        if name + ".bias" in self.sd:
            self.p[name + ".b"] = _f32(self.sd[name + ".bias"], self.dev)

    def linear(self, name, dtype=None):
        self.p[name + ".w"] = W.pack_linear(self.sd[name + ".weight"]).to(self.dev, dtype or self.dtype)
        if name + ".bias" in self.sd:
            self.p[name + ".b"] = _f32(self.sd[name + ".bias"], self.dev)

    def norm(self, name):
        self.p[name + ".g"] = _f32(self.sd[name + ".weight"], self.dev)
        self.p[name + ".b"] = _f32(self.sd[name + ".bias"], self.dev)

    def attn(self, name, self_attn, has_bias=False, qscale=None):
        """Fused [to_q; to_k] for self-attention, to_v kept separate (consumed as the A
        operand of the transposed projection); v bias folded into the out bias.
        qscale: fold this factor (head_dim^-0.5 * log2 e) into the to_q rows BEFORE the cast to the storage dtype,
        so that K Q^T is the log2-domain logit the flash kernel's prescaled loop expects (SASPA_ATTN_QPRESCALED);
        one rounding either way: bf16(c * w) here, bf16(w) and a multiply per score otherwise."""
        sd = self.sd
        if qscale is not None:
            assert not has_bias
            sd = dict(sd)
            sd[name + ".to_q.weight"] = sd[name + ".to_q.weight"].to(torch.float32) * float(qscale)
        if self_attn:
            wq, wk = sd[name + ".to_q.weight"], sd[name + ".to_k.weight"]
            self.p[name + ".qk.w"] = torch.cat([wq, wk], 0).contiguous().to(self.dev, self.dtype)
            if has_bias:
                self.p[name + ".qk.b"] = _f32(torch.cat([sd[name + ".to_q.bias"], sd[name + ".to_k.bias"]]), self.dev)
        else:
            self.p[name + ".q.w"] = sd[name + ".to_q.weight"].contiguous().to(self.dev, self.dtype)
            self.p[name + ".k.w"] = W.pack_linear(sd[name + ".to_k.weight"]).to(self.dev, self.dtype)
        self.p[name + ".v.w"] = W.pack_linear(sd[name + ".to_v.weight"]).to(self.dev, self.dtype)
        wo = sd[name + ".to_out.0.weight"]
        bo = sd[name + ".to_out.0.bias"]
        if has_bias:
            bo = bo + wo @ sd[name + ".to_v.bias"]
        self.p[name + ".o.w"] = wo.contiguous().to(self.dev, self.dtype)
        self.p[name + ".o.b"] = _f32(bo, self.dev)


# ------------------------------------------------------------------------------------------
# attention (shared by UNet / ControlNet / VAE / CLIP)
# ------------------------------------------------------------------------------------------
def xattn_enabled():
    """SASPA_XATTN=0: the cross-attention half of the level-0 blocks runs as three launches again (A/B knob)."""
    return os.environ.get("SASPA_XATTN", "1") != "0"


def ff_block_enabled():
    """The feed-forward half of the level-0 transformer blocks (LayerNorm -> GEGLU projection -> output projection -> residual) runs
    as ONE launch of saspa_ff_block (+0.3 ... 0.5 % images/s at 512x512, +1.0 ... 1.2 % at 512x704, same box, alternating:
    profiles/r6_ff_block_e2e_ab.txt).  SASPA_FF_BLOCK=0: the GEGLU launch + the output-projection launch again (A/B knob)."""
    return os.environ.get("SASPA_FF_BLOCK", "1") != "0"


def ff_block_takes(rows):
    """Is the one-launch feed-forward the faster form for this many token rows?  Its grid is one workgroup per 128 rows and one workgroup
    per CU at a time, so its time goes in steps of whole rounds of 256 workgroups (tools/ff_bench.py m=...: 24 576 rows 88 vs 108 us for
    the two launches, 32 768: 100 vs 135, 65 536: 189 vs 204, 90 112: 274 vs 315; but 16 384: 86 vs 77 and 49 152 -- one and a half
    rounds -- 176 vs 167): taken from 192 workgroups when they fill their last round to 80 %.  SASPA_FF_BLOCK_MIN_ROWS=<n> replaces the
    rule by rows >= n (tests: 0 = every size the kernel can run)."""
    mr = os.environ.get("SASPA_FF_BLOCK_MIN_ROWS")
    if mr is not None:
        return rows >= int(mr)
    wgs = rows // 128
    if wgs < 192:
        return False
    rounds = -(-wgs // 256)
    return wgs <= 256 or wgs >= 0.8 * rounds * 256


def project_vt(x, wv, nk):
    """vt[b] = Wv @ x_b^T -> [B, C, ld] with keys contiguous (pad columns zero)."""
    b, n, k = x.shape
    c = wv.shape[0]
    ld = ops.round8(nk)
    # pad columns (keys >= nk) must be zero for the unfused PV GEMM; none exist when nk % 8 == 0
    vt = (torch.zeros if ld != nk else torch.empty)((b, c, ld), device=x.device, dtype=x.dtype)
    ops.gemm_batched(wv, wv.stride(0), (0, 0), x, x.stride(1), (x.stride(0), 0), vt, ld, (c * ld, 0), c, nk, k, b, 1)
    return vt


ATTN_LOG2E = 1.4426950408889634


def rowmajor_v_enabled():
    """SASPA_ATTN_VROW=0: self-attention goes back to the separate transposed value projection (A/B knob)."""
    import os
    return os.environ.get("SASPA_ATTN_VROW", "1") != "0"


def attention_core(q, k, vt, heads, nq, nk, causal=False, prescaled=False, v_rowmajor=False):
    """q: [B,nq,C] view, k: [B,nk,C] view, vt: [B,C,ld] (v_rowmajor: V itself, [B,nk,C] view).  bf16 with head dim <= 160: fused
    flash kernel; otherwise (fp32 parity mode, 512-wide VAE head): scores GEMM -> row softmax
    -> PV GEMM, all batched over (batch, head).
    prescaled: q was projected with head_dim^-0.5 * log2(e) folded into its weights (_Packed.attn(qscale=...));
    only the flash path takes such queries."""
    b = q.shape[0]
    c = vt.shape[2] if v_rowmajor else vt.shape[1]
    d = c // heads
    out = torch.empty((b, nq, c), device=q.device, dtype=q.dtype)
    scale = d ** -0.5
    if q.dtype == torch.bfloat16 and d <= 160:
        return ops.flash_attn(q, k, vt, out, heads, d, nq, nk, scale, causal, prescaled=prescaled, v_rowmajor=v_rowmajor)
    assert not prescaled and not v_rowmajor, "prescaled queries / row-major V exist for the flash path only"
    lds = ops.round8(nk)
    assert vt.stride(1) >= lds
    scores = torch.empty((b, heads, nq, lds), device=q.device, dtype=q.dtype)
    ops.gemm_batched(q, q.stride(1), (q.stride(0), d), k, k.stride(1), (k.stride(0), d), scores, lds,
                     (heads * nq * lds, nq * lds), nq, nk, d, b, heads)
    ops.softmax_rows(scores, nk, scale, causal, nq)
    ops.gemm_batched(scores, lds, (heads * nq * lds, nq * lds), vt, vt.stride(1), (vt.stride(0), d * vt.stride(1)), out, c,
                     (nq * c, d), nq, d, lds, b, heads)
    return out


# ------------------------------------------------------------------------------------------
# UNet / ControlNet building blocks
# ------------------------------------------------------------------------------------------
class _Net:
    def __init__(self, sd, cfg, dev, dtype, fp8=False):
        self.cfg, self.dev, self.dtype = cfg, dev, dtype
        # fp8: the LayerNorm-fed projections of the transformer blocks (attn2.to_q, ff.net.0.proj) run W8A8 on
        # saspa_gemm_fp8 (bf16 networks only; blocks whose width is not a multiple of 128 stay bf16)
        self.fp8 = bool(fp8) and dtype == torch.bfloat16
        self.fp8_blocks = set()
        self.fp8_qkv = set()              # blocks whose fused self-attention projection runs on fp8 tiles (round 6)
        self.fp8_ffout = {}               # block -> calibrated?  (feed-forward output projection on fp8 tiles, round 6)
        self.xattn_blocks = set()         # transformer blocks whose cross-attention half can run as one launch (ops.xattn_block)
        self.ff_blocks = set()            # transformer blocks whose feed-forward half can run as one launch (ops.ff_block)
        self.pk = _Packed(sd, dev, dtype)
        self.p = self.pk.p
        self.temb_tables = {}
        # GroupNorm statistics out of the producers' epilogues (ops.conv(..., gn_unit=...)): every GroupNorm of the network
        # has groups of a multiple of block_out[0] / groups channels (10 for SD-1.5 / SDXL), also across a skip concat
        self.gn_unit = cfg["block_out"][0] // cfg["groups"] if dtype == torch.bfloat16 else None

    # ---- packing helpers ----
    def _pack_resnet(self, pfx, split=None):
        pk = self.pk
        pk.norm(pfx + ".norm1")
        pk.conv(pfx + ".conv1")
        pk.norm(pfx + ".norm2")
        pk.conv(pfx + ".conv2")
        if pfx + ".conv_shortcut.weight" in pk.sd:
            pk.conv(pfx + ".conv_shortcut", split)
        if pfx + ".time_emb_proj.weight" in pk.sd:
            pk.linear(pfx + ".time_emb_proj", torch.float32)
            self.resnets_with_temb.append(pfx)

    def _lvl(self, i):
        """(heads, depth) of level i: SD-1.5 = (8, 1) everywhere, SDXL per-level tuples (config.SDXL_UNET)."""
        cfg = self.cfg
        heads = cfg["heads"][i] if isinstance(cfg["heads"], (tuple, list)) else cfg["heads"]
        return heads, (cfg["depth"][i] if "depth" in cfg else 1)

    def _pack_transformer(self, pfx, lvl):
        pk = self.pk
        heads, depth = self._lvl(lvl)
        pk.norm(pfx + ".norm")
        pk.conv(pfx + ".proj_in")
        for d in range(depth):
            t = f"{pfx}.transformer_blocks.{d}"
            for n in ("norm1", "norm2", "norm3"):
                pk.norm(f"{t}.{n}")
            # bf16 networks with flash-sized heads: softmax scale * log2(e) folded into the to_q rows
            c = pk.sd[t + ".norm1.weight"].numel()
            qs = (c // heads) ** -0.5 * ATTN_LOG2E if (self.dtype == torch.bfloat16 and c // heads <= 160) else None
            pk.attn(t + ".attn1", True, qscale=qs)
            pk.attn(t + ".attn2", False, qscale=qs)
            if qs is not None:
                self.qscaled[t] = qs
            if qs is not None and c == W.XATTN_C and heads == W.XATTN_HEADS and not self.fp8 and (t + ".attn2.to_q.bias") not in pk.sd:
                # level-0 blocks: norm2 -> attn2.to_q -> attention over the text keys -> attn2.to_out + residual as ONE launch
                # (saspa_xattn_block); the stacked [to_q' ; to_out'] matrix in the order its MFMA chain consumes (weights.pack_xattn_w)
                xw, xb = W.pack_xattn_w(pk.sd[t + ".attn2.to_q.weight"].float() * float(qs), pk.sd[t + ".attn2.to_out.0.weight"].float(),
                                        pk.sd[t + ".attn2.to_out.0.bias"].float())
                self.p[t + ".attn2.xw"], self.p[t + ".attn2.xb"] = xw.to(self.dev, self.dtype), _f32(xb, self.dev)
                self.xattn_blocks.add(t)
            if self.dtype == torch.bfloat16 and c // heads <= 160 and (t + ".attn1.qk.b") not in self.p:
                # [to_q; to_k; to_v] in one matrix: at level 0 for the A-stationary kernel (LayerNorm fused, V^T written
                # transposed by the same launch: ops.linear(ln=, out_t=)), elsewhere for ONE projection launch whose V
                # columns the flash kernel reads row-major (SASPA_ATTN_V_ROWMAJOR) -- no transposed value projection
                wqk, wv = self.p[t + ".attn1.qk.w"], self.p[t + ".attn1.v.w"]
                assert wqk.shape == (2 * c, c) and wv.shape[0] == c and wv.shape[1] >= c, (wqk.shape, wv.shape)
                wqkv = torch.cat([wqk, wv[:, :c]], 0).contiguous()
                self.p[t + ".attn1.qkv.w"] = wqkv
                if wv.shape[1] == c:
                    # ONE device copy: the separate Q | K and V matrices (SASPA_ATTN_VROW=0 / the non-fused fallback) are row
                    # views of the fused buffer (0.9 GB less for the 90 SDXL blocks)
                    self.p[t + ".attn1.qk.w"], self.p[t + ".attn1.v.w"] = wqkv[:2 * c], wqkv[2 * c:]
            packed = W.pack_geglu(pk.sd[t + ".ff.net.0.proj.weight"], pk.sd[t + ".ff.net.0.proj.bias"]) \
                if self.dtype == torch.bfloat16 else None
            if packed is not None:          # bf16: GEGLU fused into the projection's epilogue
                self.p[t + ".ff.net.0.proj.w"] = packed[0].to(self.dev, self.dtype)
                self.p[t + ".ff.net.0.proj.b"] = _f32(packed[1], self.dev)
                self.fused_geglu.add(t)
            else:
                pk.linear(t + ".ff.net.0.proj")
            pk.linear(t + ".ff.net.2")
            if packed is not None and c == 320 and not self.fp8 and ff_block_enabled():
                # level 0: the whole feed-forward as one launch -- W1 per 32-feature slice [values | gates], W2 as the MFMA fragments
                # stage C consumes (weights.pack_ff_block); the two-launch operands stay (tokens the kernel does not take)
                fw1, fb1, fw2, fb2 = W.pack_ff_block(pk.sd[t + ".ff.net.0.proj.weight"], pk.sd[t + ".ff.net.0.proj.bias"],
                                                     pk.sd[t + ".ff.net.2.weight"], pk.sd[t + ".ff.net.2.bias"])
                self.p[t + ".ffb.w1"], self.p[t + ".ffb.b1"] = fw1.to(self.dev, self.dtype), _f32(fb1, self.dev)
                self.p[t + ".ffb.w2f"], self.p[t + ".ffb.b2"] = fw2.to(self.dev, self.dtype), _f32(fb2, self.dev)
                self.ff_blocks.add(t)
            if self.fp8 and pk.sd[t + ".norm2.weight"].numel() % 128 == 0:
                self._quantize_block(t)
            self.blocks.append(t)
        pk.conv(pfx + ".proj_out")
        self.tr_info[pfx] = (heads, depth)

    def _pack_encoder(self):
        cfg, pk = self.cfg, self.pk
        self.resnets_with_temb, self.blocks, self.tr_info, self.fused_geglu, self.qscaled = [], [], {}, set(), {}
        pk.conv("conv_in")
        pk.linear("time_embedding.linear_1", torch.float32)
        pk.linear("time_embedding.linear_2", torch.float32)
        if "add_embed" in cfg:
            pk.linear("add_embedding.linear_1", torch.float32)
            pk.linear("add_embedding.linear_2", torch.float32)
        n_lvl = len(cfg["block_out"])
        for i in range(n_lvl):
            for j in range(cfg["layers"]):
                self._pack_resnet(f"down_blocks.{i}.resnets.{j}")
                if cfg["attn"][i]:
                    self._pack_transformer(f"down_blocks.{i}.attentions.{j}", i)
            if i != n_lvl - 1:
                pk.conv(f"down_blocks.{i}.downsamplers.0.conv")
        self._pack_resnet("mid_block.resnets.0")
        self._pack_transformer("mid_block.attentions.0", n_lvl - 1)
        self._pack_resnet("mid_block.resnets.1")

    # ---- hoisted, time-invariant state ----
    @staticmethod
    def _sinusoid(values, dim):
        """Timesteps(dim, flip_sin_to_cos=True, freq_shift=0) of a 1-D list of values -> [n, dim] fp32 (host)."""
        half = dim // 2
        t = torch.as_tensor(values, dtype=torch.float32).reshape(-1)
        exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
        ang = t[:, None] * torch.exp(exponent)[None, :]
        return torch.cat([torch.cos(ang), torch.sin(ang)], -1)

    def prepare_timesteps(self, timesteps, added=None):
        """Time-embedding MLP and every resnet's time_emb_proj for ALL steps (fp32 GEMMs):
        table[pfx] = [steps, Cout].  The sinusoid is host scheduler state, like the DDIM
        coefficients (Timesteps(dim0, flip_sin_to_cos=True, freq_shift=0)).
        SDXL (`add_embed` in the config): `added` = (text_embeds [B, pooled] device fp32, time_ids [B, 6] host) adds
        the per-sample text_time embedding, so the tables become [steps, B, Cout] (row vector per batch sample)."""
        sinus = ops.h2d(self._sinusoid(timesteps, self.cfg["block_out"][0]), self.dev)
        p = self.p
        e = ops.linear(sinus, p["time_embedding.linear_1.w"], p["time_embedding.linear_1.b"], act=SILU)
        e = ops.linear(e, p["time_embedding.linear_2.w"], p["time_embedding.linear_2.b"])       # [S, temb]
        if "add_embed" in self.cfg:
            if added is None:
                raise ValueError("this network needs the SDXL added conditioning (text_embeds, time_ids)")
            text_embeds, time_ids = added
            ae = self.cfg["add_embed"]
            b = text_embeds.shape[0]
            te = ops.h2d(self._sinusoid(torch.as_tensor(time_ids, dtype=torch.float32), ae["time_dim"]).reshape(b, -1), self.dev)
            a_in = torch.cat([text_embeds.to(torch.float32), te], -1).contiguous()
            a = ops.linear(a_in, p["add_embedding.linear_1.w"], p["add_embedding.linear_1.b"], act=SILU)
            # emb[s][b] = time_emb[s] + aug[b]: the aug row block is the GEMM's residual, once per step
            s_n = e.shape[0]
            e_rep = e[:, None, :].expand(s_n, b, e.shape[1]).reshape(s_n * b, -1).contiguous()
            a_rep = a[None].expand(s_n, b, a.shape[1]).reshape(s_n * b, -1).contiguous()
            e = ops.linear(a_rep, p["add_embedding.linear_2.w"], p["add_embedding.linear_2.b"], residual=e_rep)
            shape = (s_n, b, -1)
        else:
            shape = (e.shape[0], -1)
        se = ops.activation(e, SILU)
        # every resnet's time_emb_proj as ONE GEMM against the row-concatenated weights: table [steps(, B), total];
        # resnet pfx owns the column block [off, off + Cout)
        if "temb_cat.w" not in p:
            p["temb_cat.w"] = torch.cat([p[pfx + ".time_emb_proj.w"] for pfx in self.resnets_with_temb], 0).contiguous()
            p["temb_cat.b"] = torch.cat([p[pfx + ".time_emb_proj.b"] for pfx in self.resnets_with_temb], 0).contiguous()
            self.temb_off, off = {}, 0
            for pfx in self.resnets_with_temb:
                c = p[pfx + ".time_emb_proj.w"].shape[0]
                self.temb_off[pfx] = (off, c)
                off += c
            self.temb_total = off
            for pfx in self.resnets_with_temb:                  # the per-resnet copies are no longer needed
                del p[pfx + ".time_emb_proj.w"], p[pfx + ".time_emb_proj.b"]
        tab = ops.linear(se, p["temb_cat.w"], p["temb_cat.b"])
        self.temb_all = tab[:, :self.temb_total].contiguous().view(*shape[:-1], self.temb_total)
        self.temb_tables = {pfx: self.temb_all[..., o:o + c] for pfx, (o, c) in self.temb_off.items()}

    def bind_step_state(self, cur):
        """hipGraph replays: `cur` ([total] or [B, total] fp32, one row of `temb_all`, refreshed on the device by
        ops.gather_row) is what the resnets' row vectors point into when they are called with step=None."""
        self.temb_cur = cur
        self.temb_cur_views = {pfx: cur[..., o:o + c] for pfx, (o, c) in self.temb_off.items()}

    def prepare_context(self, ctx):
        """Cross-attention K and V^T of every transformer block for a [B,77,ctx_dim] batch."""
        b, n, _ = ctx.shape
        self.ctx_kv = {}
        for t in self.blocks:
            a = t + ".attn2"
            k = ops.linear(ctx, self.p[a + ".k.w"])                     # [B,77,C]
            vt = project_vt(ctx, self.p[a + ".v.w"], n)                 # [B,C,80]
            if t in self.xattn_blocks and xattn_enabled() and n <= 96:    # (longer contexts: the three-launch path of transformer())
                # the same K / V cut into the MFMA operand fragments of saspa_xattn_block (time-invariant, like K / V^T themselves)
                kf, vf = W.xattn_kv_fragments(k, vt[:, :, :n].transpose(1, 2).contiguous())
                self.ctx_kv[t] = (k, vt, n, kf, vf)
            else:
                self.ctx_kv[t] = (k, vt, n)

    # ---- blocks ----
    def resnet(self, pfx, x, step, eps, x2=None):
        p, g = self.p, self.cfg["groups"]
        rv = None
        if pfx in self.temb_tables:
            rv = self.temb_cur_views[pfx] if step is None else self.temb_tables[pfx][step]
        # (round 6: the halo-tiled conv with the GroupNorm applied in LDS -- saspa_conv3x3_halo, SASPA_HALO=1 in round 5 -- is no longer
        # reachable from the pipeline: parity-green, -1.4 % end to end, both arms in profiles/EXPERIMENTS.md; the kernel stays a tested
        # library entry point)
        h = ops.groupnorm(x, p[pfx + ".norm1.g"], p[pfx + ".norm1.b"], g, eps, SILU, x2=x2)
        # conv1 -> norm2 -> SiLU: conv1's output has no other reader (fuse_gn: one launch for reduce + GroupNorm at the small levels)
        h = ops.conv(h, p[pfx + ".conv1.w"], p[pfx + ".conv1.b"], kh=3, kw=3, pad=1, rowvec=rv, gn_unit=self.gn_unit,
                     fuse_gn=(p[pfx + ".norm2.g"], p[pfx + ".norm2.b"], g, eps, SILU))
        # (the 1x1 shortcut conv on the side stream beside norm1 / conv1 -- MFMA-bound next to an HBM-bound pass -- measured +-0.1 % at
        # 512x512 / 512x704 / 512x768 in round 6, like round 4's finer forks: profiles/r6_sc_fork_ab.txt; not kept)
        if pfx + ".conv_shortcut.w" in p:
            sc = ops.conv(x, p[pfx + ".conv_shortcut.w"], p[pfx + ".conv_shortcut.b"], x2=x2)
        else:
            assert x2 is None
            sc = x
        return ops.conv(h, p[pfx + ".conv2.w"], p[pfx + ".conv2.b"], kh=3, kw=3, pad=1, residual=sc, gn_unit=self.gn_unit)

    def _quantize_block(self, t):
        """e4m3 copies (per-output-channel scales) of a transformer block's projections for saspa_gemm_fp8.
        Round 3: the two projections that read a LayerNorm's output through `saspa_layernorm_quant_fp8` (attn2.to_q, ff.net.0.proj;
        the GEGLU projection's rows regrouped for the fp8 kernel's 128-column tiles).  Round 6 (SASPA_FP8_BREADTH=all, the
        default; =ln restores round 3): also the fused self-attention [to_q; to_k; to_v] -- same LayerNorm-quantise pass, the
        flash kernel reads its V columns row-major -- and the feed-forward OUTPUT projection ff.net.2, whose input leaves the
        GEGLU epilogue as e4m3 under one calibrated tensor-wide power-of-two scale (`ff.amax` / `ff.scale`, see transformer()):
        16 of the 18 C^2 multiply-accumulates per token of a block run on fp8 tiles (attn1 / attn2 to_out stay bf16: their input
        is the attention kernel's output)."""
        sd, p = self.pk.sd, self.p
        wq, sw = W.quantize_fp8(sd[t + ".attn2.to_q.weight"].float() * self.qscaled.get(t, 1.0))   # same fold as the bf16 rows
        p[t + ".attn2.q.w8"], p[t + ".attn2.q.sw"] = wq.to(self.dev), sw.to(self.dev)
        wg, bg = W.pack_geglu_tile(sd[t + ".ff.net.0.proj.weight"].float(), sd[t + ".ff.net.0.proj.bias"].float(), 128)
        wq, sw = W.quantize_fp8(wg)
        p[t + ".ff.net.0.proj.w8"], p[t + ".ff.net.0.proj.sw"], p[t + ".ff.net.0.proj.b8"] = wq.to(self.dev), sw.to(self.dev), _f32(bg, self.dev)
        del p[t + ".attn2.q.w"], p[t + ".ff.net.0.proj.w"], p[t + ".ff.net.0.proj.b"]
        self.fp8_blocks.add(t)
        if os.environ.get("SASPA_FP8_BREADTH", "all") != "all":
            return
        c = sd[t + ".norm1.weight"].numel()
        if (t + ".attn1.qkv.w") in p and rowmajor_v_enabled() and (t + ".attn1.to_q.bias") not in sd:
            wqkv = torch.cat([sd[t + ".attn1.to_q.weight"].float() * self.qscaled.get(t, 1.0), sd[t + ".attn1.to_k.weight"].float(),
                              sd[t + ".attn1.to_v.weight"].float()], 0)
            assert wqkv.shape == (3 * c, c)
            wq, sw = W.quantize_fp8(wqkv)
            p[t + ".attn1.qkv.w8"], p[t + ".attn1.qkv.sw"] = wq.to(self.dev), sw.to(self.dev)
            for k in (".attn1.qkv.w", ".attn1.qk.w", ".attn1.v.w"):
                p.pop(t + k, None)
            self.fp8_qkv.add(t)
        if (4 * c) % 128 == 0:
            wq, sw = W.quantize_fp8(sd[t + ".ff.net.2.weight"].float())
            p[t + ".ff.net.2.w8"], p[t + ".ff.net.2.sw"] = wq.to(self.dev), sw.to(self.dev)
            del p[t + ".ff.net.2.w"]
            p[t + ".ff.amax"] = torch.zeros(1, device=self.dev, dtype=torch.float32)
            p[t + ".ff.scale"] = torch.ones(1, device=self.dev, dtype=torch.float32)
            self.fp8_ffout[t] = False          # -> True once the scale has been calibrated (first execution of the block)

    def transformer(self, pfx, x):
        p, g = self.p, self.cfg["groups"]
        heads, depth = self.tr_info[pfx]
        b, hh, ww, c = x.shape
        n = hh * ww
        h = ops.groupnorm(x, p[pfx + ".norm.g"], p[pfx + ".norm.b"], g, 1e-6)
        h = ops.conv(h, p[pfx + ".proj_in.w"], p[pfx + ".proj_in.b"]).view(b, n, c)
        for d in range(depth):
            t = f"{pfx}.transformer_blocks.{d}"
            # self-attention.  Level 0 of a full-size batch: LayerNorm + Q | K + V^T in ONE launch of the A-stationary kernel
            # (saspa_gemm_as.hip) instead of three launches that each re-read the tokens
            wqkv = p.get(t + ".attn1.qkv.w")
            fuse = wqkv is not None and c == 320 and t not in self.fp8_blocks and n % 32 == 0 and \
                ops.linear_ln_fusable(h, wqkv, n_out=2 * c)
            pre = t in self.qscaled
            if t in self.fp8_qkv:
                # W8A8 (round 6): LayerNorm + per-token quantisation in one pass, ONE e4m3 projection launch for Q | K | V
                q8, s8 = ops.layernorm_quant_fp8(h, p[t + ".norm1.g"], p[t + ".norm1.b"])
                qkv = ops.linear_fp8(q8, s8, p[t + ".attn1.qkv.w8"], p[t + ".attn1.qkv.sw"])          # [B,N,3C] bf16
                o = attention_core(qkv[:, :, :c], qkv[:, :, c:2 * c], qkv[:, :, 2 * c:3 * c], heads, n, n, prescaled=pre, v_rowmajor=True)
            elif fuse:
                vt = torch.empty((b, c, n), device=h.device, dtype=h.dtype)
                qk = ops.linear(h, wqkv, None, ln=(p[t + ".norm1.g"], p[t + ".norm1.b"], 1e-5), out_t=vt, n_split=2 * c, rows_per_batch=n)
                o = attention_core(qk[:, :, :c], qk[:, :, c:], vt, heads, n, n, prescaled=pre)
            elif wqkv is not None and rowmajor_v_enabled():
                # one projection launch; the attention kernel takes its V columns as they are
                n1 = ops.layernorm(h, p[t + ".norm1.g"], p[t + ".norm1.b"])
                qkv = ops.linear(n1, wqkv)                                     # [B,N,3C]
                o = attention_core(qkv[:, :, :c], qkv[:, :, c:2 * c], qkv[:, :, 2 * c:3 * c], heads, n, n, prescaled=pre, v_rowmajor=True)
            else:
                n1 = ops.layernorm(h, p[t + ".norm1.g"], p[t + ".norm1.b"])
                qk = ops.linear(n1, p[t + ".attn1.qk.w"])                     # [B,N,2C]
                vt = project_vt(n1, p[t + ".attn1.v.w"], n)
                o = attention_core(qk[:, :, :c], qk[:, :, c:], vt, heads, n, n, prescaled=pre)
            h = ops.linear(o, p[t + ".attn1.o.w"], p[t + ".attn1.o.b"], residual=h)
            if t in self.fp8_blocks:
                # W8A8: LayerNorm + per-token quantisation in one pass, e4m3 x e4m3 MFMA, scales applied in the epilogue
                q8, s8 = ops.layernorm_quant_fp8(h, p[t + ".norm2.g"], p[t + ".norm2.b"])
                q = ops.linear_fp8(q8, s8, p[t + ".attn2.q.w8"], p[t + ".attn2.q.sw"])
                k, vtc, nk = self.ctx_kv[t][:3]
                o = attention_core(q, k, vtc, heads, n, nk, prescaled=pre)
                h = ops.linear(o, p[t + ".attn2.o.w"], p[t + ".attn2.o.b"], residual=h)
                q8, s8 = ops.layernorm_quant_fp8(h, p[t + ".norm3.g"], p[t + ".norm3.b"])
                if t in self.fp8_ffout:
                    # the hidden state leaves the GEGLU epilogue as e4m3 under ONE tensor-wide scale and the output projection
                    # reads the bytes.  The scale is calibrated the first time the block runs (always an eager launch: a step graph
                    # is captured after an eager warm-up step): one extra GEGLU launch measures max |value|, the scale becomes the
                    # power of two >= 16 * max / 448 -- e4m3 is a floating format, a power-of-two scale only shifts exponents, so
                    # the bytes do not depend on the calibration batch short of overflow (4 binades of head room, saturating)
                    if not self.fp8_ffout[t]:
                        if torch.cuda.is_current_stream_capturing():
                            raise RuntimeError("fp8 feed-forward scale of %s is not calibrated: run one eager evaluation before capture" % t)
                        ops.linear_fp8(q8, s8, p[t + ".ff.net.0.proj.w8"], p[t + ".ff.net.0.proj.sw"], p[t + ".ff.net.0.proj.b8"],
                                       act=ops.ACT_GEGLU, amax=p[t + ".ff.amax"])
                        p[t + ".ff.scale"].copy_(ops.fp8_pow2_scale(p[t + ".ff.amax"]))
                        self.fp8_ffout[t] = True
                    ff8 = ops.linear_fp8(q8, s8, p[t + ".ff.net.0.proj.w8"], p[t + ".ff.net.0.proj.sw"], p[t + ".ff.net.0.proj.b8"],
                                         act=ops.ACT_GEGLU, out_fp8_scale=p[t + ".ff.scale"])
                    h = ops.linear_fp8(ff8, p[t + ".ff.scale"], p[t + ".ff.net.2.w8"], p[t + ".ff.net.2.sw"], p[t + ".ff.net.2.b"], residual=h)
                    continue
                ff = ops.linear_fp8(q8, s8, p[t + ".ff.net.0.proj.w8"], p[t + ".ff.net.0.proj.sw"], p[t + ".ff.net.0.proj.b8"],
                                    act=ops.ACT_GEGLU)
                h = ops.linear(ff, p[t + ".ff.net.2.w"], p[t + ".ff.net.2.b"], residual=h)
                continue
            # cross-attention against the cached text K / V^T
            kv = self.ctx_kv[t]
            if fuse and len(kv) == 5 and n % 256 == 0 and kv[2] <= 96:
                # level 0 of a full-size batch: LayerNorm + to_q + attention + to_out + residual in one launch
                h = ops.xattn_block(h, (p[t + ".norm2.g"], p[t + ".norm2.b"], 1e-5), p[t + ".attn2.xw"], p[t + ".attn2.xb"], kv[3], kv[4],
                                    kv[2], n)
            else:
                if fuse and ops.linear_ln_fusable(h, p[t + ".attn2.q.w"]):
                    q = ops.linear(h, p[t + ".attn2.q.w"], ln=(p[t + ".norm2.g"], p[t + ".norm2.b"], 1e-5))
                else:
                    n2 = ops.layernorm(h, p[t + ".norm2.g"], p[t + ".norm2.b"])
                    q = ops.linear(n2, p[t + ".attn2.q.w"])
                k, vtc, nk = kv[:3]
                o = attention_core(q, k, vtc, heads, n, nk, prescaled=pre)
                h = ops.linear(o, p[t + ".attn2.o.w"], p[t + ".attn2.o.b"], residual=h)
            # GEGLU feed-forward
            # (with a ragged last round of row blocks -- 512x704 -- the wave-specialised kernel + LayerNorm is as fast: == 2)
            if t in self.ff_blocks and ff_block_takes(b * n) and ops.ff_block_eligible(h, p[t + ".ffb.w1"], p[t + ".ffb.w2f"]):
                h = ops.ff_block(h, (p[t + ".norm3.g"], p[t + ".norm3.b"], 1e-5), p[t + ".ffb.w1"], p[t + ".ffb.b1"], p[t + ".ffb.w2f"],
                                 p[t + ".ffb.b2"], residual=h)
                continue
            if fuse and t in self.fused_geglu and ops.linear_ln_fusable(h, p[t + ".ff.net.0.proj.w"], act=ops.ACT_GEGLU) == 2:
                ff = ops.linear(h, p[t + ".ff.net.0.proj.w"], p[t + ".ff.net.0.proj.b"], act=ops.ACT_GEGLU,
                                ln=(p[t + ".norm3.g"], p[t + ".norm3.b"], 1e-5))
            else:
                n3 = ops.layernorm(h, p[t + ".norm3.g"], p[t + ".norm3.b"])
                if t in self.fused_geglu:
                    ff = ops.linear(n3, p[t + ".ff.net.0.proj.w"], p[t + ".ff.net.0.proj.b"], act=ops.ACT_GEGLU)
                else:
                    ff = ops.geglu(ops.linear(n3, p[t + ".ff.net.0.proj.w"], p[t + ".ff.net.0.proj.b"]))
            h = ops.linear(ff, p[t + ".ff.net.2.w"], p[t + ".ff.net.2.b"], residual=h)
        return ops.conv(h.view(b, hh, ww, c), p[pfx + ".proj_out.w"], p[pfx + ".proj_out.b"], residual=x, gn_unit=self.gn_unit)

    def encode(self, sample, step, conv_in_residual=None):
        """conv_in + down blocks + mid block.  Returns (mid, [skips])."""
        cfg, p = self.cfg, self.p
        s = ops.conv(sample, p["conv_in.w"], p["conv_in.b"], kh=3, kw=3, pad=1, residual=conv_in_residual, gn_unit=self.gn_unit)
        skips = [s]
        n_lvl = len(cfg["block_out"])
        for i in range(n_lvl):
            for j in range(cfg["layers"]):
                s = self.resnet(f"down_blocks.{i}.resnets.{j}", s, step, 1e-5)
                if cfg["attn"][i]:
                    s = self.transformer(f"down_blocks.{i}.attentions.{j}", s)
                skips.append(s)
            if i != n_lvl - 1:
                d = f"down_blocks.{i}.downsamplers.0.conv"
                s = ops.conv(s, p[d + ".w"], p[d + ".b"], kh=3, kw=3, stride=2, pad=1, gn_unit=self.gn_unit)
                skips.append(s)
        s = self.resnet("mid_block.resnets.0", s, step, 1e-5)
        s = self.transformer("mid_block.attentions.0", s)
        s = self.resnet("mid_block.resnets.1", s, step, 1e-5)
        return s, skips


class UNet(_Net):
    """UNet2DConditionModel (SD-1.5 topology)."""

    def __init__(self, sd, cfg, dev, dtype, fp8=False):
        super().__init__(sd, cfg, dev, dtype, fp8)
        self._pack_encoder()
        pk = self.pk
        bo = cfg["block_out"]
        # skip channel bookkeeping (mirrors the encoder)
        skip_ch = [bo[0]]
        for i, c in enumerate(bo):
            skip_ch += [c] * cfg["layers"]
            if i != len(bo) - 1:
                skip_ch.append(c)
        rev = list(reversed(bo))
        prev = rev[0]
        n_lvl = len(bo)
        for i, c in enumerate(rev):
            for j in range(cfg["layers"] + 1):
                sk = skip_ch.pop()
                self._pack_resnet(f"up_blocks.{i}.resnets.{j}", split=(prev, sk))
                if list(reversed(cfg["attn"]))[i]:
                    self._pack_transformer(f"up_blocks.{i}.attentions.{j}", n_lvl - 1 - i)
                prev = c
            if i != n_lvl - 1:
                pk.conv(f"up_blocks.{i}.upsamplers.0.conv")
        pk.norm("conv_norm_out")
        pk.conv("conv_out")
        pk.sd = None  # drop the fp32 host copy reference

    def decode(self, mid, skips, step, out=None):
        cfg, p = self.cfg, self.p
        s = mid
        skips = list(skips)
        n_lvl = len(cfg["block_out"])
        rev_attn = list(reversed(cfg["attn"]))
        for i in range(n_lvl):
            for j in range(cfg["layers"] + 1):
                sk = skips.pop()
                s = self.resnet(f"up_blocks.{i}.resnets.{j}", s, step, 1e-5, x2=sk)
                if rev_attn[i]:
                    s = self.transformer(f"up_blocks.{i}.attentions.{j}", s)
            if i != n_lvl - 1:
                u = f"up_blocks.{i}.upsamplers.0.conv"
                s = ops.conv(s, p[u + ".w"], p[u + ".b"], kh=3, kw=3, pad=1, upsample=True, gn_unit=self.gn_unit)
        s = ops.groupnorm(s, p["conv_norm_out.g"], p["conv_norm_out.b"], cfg["groups"], 1e-5, SILU)
        return ops.conv(s, p["conv_out.w"], p["conv_out.b"], kh=3, kw=3, pad=1, out=out)

    def forward(self, sample, step):
        """UNet forward without ControlNet residuals."""
        mid, skips = self.encode(sample, step)
        return self.decode(mid, skips, step)


class ControlNet(_Net):
    """ControlNetModel (control_v11p_sd15_canny topology)."""

    def __init__(self, sd, cfg, dev, dtype, fp8=False):
        super().__init__(sd, cfg, dev, dtype, fp8)
        self._pack_encoder()
        pk = self.pk
        ce = cfg["cond_embed"]
        e = "controlnet_cond_embedding"
        pk.conv(e + ".conv_in")
        for i in range(2 * (len(ce) - 1)):
            pk.conv(f"{e}.blocks.{i}")
        pk.conv(e + ".conv_out")
        n_skips = 1 + sum(cfg["layers"] + (1 if i != len(cfg["block_out"]) - 1 else 0) for i in range(len(cfg["block_out"])))
        self.n_skips = n_skips
        for i in range(n_skips):
            pk.conv(f"controlnet_down_blocks.{i}")
        pk.conv("controlnet_mid_block")
        pk.sd = None

    def cond_embedding(self, cond):
        """ControlNetConditioningEmbedding on [B,H,W,8] control images in [0,1]; time-invariant,
        computed once per image (diffusers recomputes it every step)."""
        p = self.p
        e = "controlnet_cond_embedding"
        h = ops.conv(cond, p[e + ".conv_in.w"], p[e + ".conv_in.b"], kh=3, kw=3, pad=1, act=SILU)
        for i in range(len(self.cfg["cond_embed"]) - 1):
            h = ops.conv(h, p[f"{e}.blocks.{2 * i}.w"], p[f"{e}.blocks.{2 * i}.b"], kh=3, kw=3, pad=1, act=SILU)
            h = ops.conv(h, p[f"{e}.blocks.{2 * i + 1}.w"], p[f"{e}.blocks.{2 * i + 1}.b"], kh=3, kw=3, stride=2, pad=1,
                         act=SILU)
        return ops.conv(h, p[e + ".conv_out.w"], p[e + ".conv_out.b"], kh=3, kw=3, pad=1)

    def forward(self, sample, step, cond_emb, scale, unet_skips=None, unet_mid=None):
        """Returns ([12 residuals], mid residual), each scale*(zero_conv(feature)) and, when the
        UNet encoder outputs are given, already summed with them (fused epilogue)."""
        mid, feats = self.encode(sample, step, conv_in_residual=cond_emb)
        return self.zero_convs(mid, feats, scale, unet_skips, unet_mid)

    def zero_convs(self, mid, feats, scale, unet_skips=None, unet_mid=None):
        """The 12 + 1 zero convolutions on the encoder features (x conditioning scale, + the UNet's own skips / mid when
        given): the point where the ControlNet branch joins the UNet."""
        p = self.p
        outs = []
        for i, f in enumerate(feats):
            r = None if unet_skips is None else unet_skips[i]
            outs.append(ops.conv(f, p[f"controlnet_down_blocks.{i}.w"], p[f"controlnet_down_blocks.{i}.b"], alpha=scale,
                                 residual=r, gn_unit=self.gn_unit))
        m = ops.conv(mid, p["controlnet_mid_block.w"], p["controlnet_mid_block.b"], alpha=scale, residual=unet_mid,
                     gn_unit=self.gn_unit)
        return outs, m


class VAEDecoder:
    """AutoencoderKL.decode."""

    def __init__(self, sd, cfg, dev, dtype, f32_gemm="exact"):
        self.cfg, self.dev, self.dtype = cfg, dev, dtype
        # fp32 VAE only: "exact" = fp32 MFMA (parity path), "x3" = three bf16 MFMAs per product (ops.f32_gemm_mode)
        self.f32_gemm = f32_gemm if dtype == torch.float32 else "exact"
        pk = self.pk = _Packed(sd, dev, dtype)
        self.p = pk.p
        pk.conv("post_quant_conv")
        pk.conv("decoder.conv_in")
        self._pack_resnet("decoder.mid_block.resnets.0")
        a = "decoder.mid_block.attentions.0"
        pk.norm(a + ".group_norm")
        pk.attn(a, True, has_bias=True)
        self._pack_resnet("decoder.mid_block.resnets.1")
        n_lvl = len(cfg["block_out"])
        for i in range(n_lvl):
            for j in range(cfg["layers"] + 1):
                self._pack_resnet(f"decoder.up_blocks.{i}.resnets.{j}")
            if i != n_lvl - 1:
                pk.conv(f"decoder.up_blocks.{i}.upsamplers.0.conv")
        pk.norm("decoder.conv_norm_out")
        pk.conv("decoder.conv_out")
        pk.sd = None
        if self.f32_gemm == "x3" and os.environ.get("SASPA_X3_PRESPLIT", "1") != "0":
            self._presplit()

    def _presplit(self):
        """SASPA_F32X3 (round 6): the conv weights are stored PRE-SPLIT into their bf16 hi | lo halves (weights.presplit_x3,
        SaspaGemmParams.w_split): the K loop of those GEMMs is bound by the VALU work of splitting BOTH operands in registers
        (9 splits of 8 values per 60 MFMAs on a 128 x 160 tile), the weights' half of it is time-invariant.  Same products in the
        same order -- bit-identical to the in-kernel split.  Layers the LDS-DMA loader cannot take (conv_in: 8 input channels)
        keep fp32 weights.  SASPA_X3_PRESPLIT=0 turns it off (A/B)."""
        for k, t in list(self.p.items()):
            cin = getattr(t, "saspa_cin", 0)           # set by _Packed.conv: only conv weights carry it
            if not k.endswith(".w") or t.dtype != torch.float32 or cin == 0 or cin % 32 or t.shape[1] % 32:
                continue
            t2 = W.presplit_x3(t)
            t2.saspa_korder, t2.saspa_cin, t2.saspa_wsplit = t.saspa_korder, cin, 1
            self.p[k] = t2

    def _pack_resnet(self, pfx):
        pk = self.pk
        pk.norm(pfx + ".norm1")
        pk.conv(pfx + ".conv1")
        pk.norm(pfx + ".norm2")
        pk.conv(pfx + ".conv2")
        if pfx + ".conv_shortcut.weight" in pk.sd:
            pk.conv(pfx + ".conv_shortcut")

    def resnet(self, pfx, x):
        p, g = self.p, self.cfg["groups"]
        h = ops.groupnorm(x, p[pfx + ".norm1.g"], p[pfx + ".norm1.b"], g, 1e-6, SILU)
        h = ops.conv(h, p[pfx + ".conv1.w"], p[pfx + ".conv1.b"], kh=3, kw=3, pad=1)
        h = ops.groupnorm(h, p[pfx + ".norm2.g"], p[pfx + ".norm2.b"], g, 1e-6, SILU)
        sc = x
        if pfx + ".conv_shortcut.w" in p:
            sc = ops.conv(x, p[pfx + ".conv_shortcut.w"], p[pfx + ".conv_shortcut.b"])
        return ops.conv(h, p[pfx + ".conv2.w"], p[pfx + ".conv2.b"], kh=3, kw=3, pad=1, residual=sc)

    def mid_attention(self, x):
        return self._mid_attention(x, "decoder.mid_block.attentions.0")

    def _mid_attention(self, x, a):
        p = self.p
        b, hh, ww, c = x.shape
        n = hh * ww
        h = ops.groupnorm(x, p[a + ".group_norm.g"], p[a + ".group_norm.b"], self.cfg["groups"], 1e-6).view(b, n, c)
        qk = ops.linear(h, p[a + ".qk.w"], p[a + ".qk.b"])
        vt = project_vt(h, p[a + ".v.w"], n)
        o = attention_core(qk[:, :, :c], qk[:, :, c:], vt, 1, n, n)
        return ops.linear(o, p[a + ".o.w"], p[a + ".o.b"], residual=x.view(b, n, c)).view(b, hh, ww, c)

    def decode(self, z):
        """z: [B,h,w,8] latents ALREADY divided by the scaling factor -> [B,8h,8w,8] (3 live).  The kernels address
        an operand with 32-bit byte offsets (< 2 GiB): the widest full-resolution activation (block_out[1] channels)
        bounds the images per launch sequence -- 8 at 512x512 in bf16, 3 in fp32 (upcast VAE), 1 at 1024x1024 fp32."""
        b, h, w, _ = z.shape
        esz = 2 if self.dtype == torch.bfloat16 else 4
        per_img = 64 * h * w * self.cfg["block_out"][min(1, len(self.cfg["block_out"]) - 1)] * esz
        chunk = max(1, ((1 << 31) - 1) // per_img)
        with ops.f32_gemm_mode(self.f32_gemm):
            if b <= chunk:
                return self._decode(z)
            return torch.cat([self._decode(z[i:i + chunk].contiguous()) for i in range(0, b, chunk)], 0)

    def _decode(self, z):
        p, cfg = self.p, self.cfg
        h = ops.conv(z, p["post_quant_conv.w"], p["post_quant_conv.b"])
        h = ops.conv(h, p["decoder.conv_in.w"], p["decoder.conv_in.b"], kh=3, kw=3, pad=1)
        h = self.resnet("decoder.mid_block.resnets.0", h)
        h = self.mid_attention(h)
        h = self.resnet("decoder.mid_block.resnets.1", h)
        n_lvl = len(cfg["block_out"])
        for i in range(n_lvl):
            for j in range(cfg["layers"] + 1):
                h = self.resnet(f"decoder.up_blocks.{i}.resnets.{j}", h)
            if i != n_lvl - 1:
                u = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                h = ops.conv(h, p[u + ".w"], p[u + ".b"], kh=3, kw=3, pad=1, upsample=True)
        h = ops.groupnorm(h, p["decoder.conv_norm_out.g"], p["decoder.conv_norm_out.b"], cfg["groups"], 1e-6, SILU)
        return ops.conv(h, p["decoder.conv_out.w"], p["decoder.conv_out.b"], kh=3, kw=3, pad=1)


class VAEEncoder(VAEDecoder):
    """AutoencoderKL.encode up to the latent distribution's parameters (SDEdit / img2img): conv_in, down
    blocks (2 resnets + a stride-2 conv that zero-pads right / bottom only), mid block (resnet, single-head attention,
    resnet), GroupNorm + SiLU, conv_out, quant_conv -> [B,h,w,8] = mean (channels 0..3) | logvar (4..7)."""

    def __init__(self, sd, cfg, dev, dtype):
        self.cfg, self.dev, self.dtype = cfg, dev, dtype
        pk = self.pk = _Packed(sd, dev, dtype)
        self.p = pk.p
        pk.conv("encoder.conv_in")
        n_lvl = len(cfg["block_out"])
        for i in range(n_lvl):
            for j in range(cfg["layers"]):
                self._pack_resnet(f"encoder.down_blocks.{i}.resnets.{j}")
            if i != n_lvl - 1:
                pk.conv(f"encoder.down_blocks.{i}.downsamplers.0.conv")
        self._pack_resnet("encoder.mid_block.resnets.0")
        a = "encoder.mid_block.attentions.0"
        pk.norm(a + ".group_norm")
        pk.attn(a, True, has_bias=True)
        self._pack_resnet("encoder.mid_block.resnets.1")
        pk.norm("encoder.conv_norm_out")
        pk.conv("encoder.conv_out")
        pk.conv("quant_conv")
        pk.sd = None

    def encode(self, x):
        """x: [B,H,W,8] pixels in [-1,1] (3 live channels) -> moments [B,H/8,W/8,8]."""
        p, cfg = self.p, self.cfg
        h = ops.conv(x, p["encoder.conv_in.w"], p["encoder.conv_in.b"], kh=3, kw=3, pad=1)
        n_lvl = len(cfg["block_out"])
        for i in range(n_lvl):
            for j in range(cfg["layers"]):
                h = self.resnet(f"encoder.down_blocks.{i}.resnets.{j}", h)
            if i != n_lvl - 1:
                d = f"encoder.down_blocks.{i}.downsamplers.0.conv"
                h = ops.conv(h, p[d + ".w"], p[d + ".b"], kh=3, kw=3, stride=2, pad=0, out_hw=(h.shape[1] // 2, h.shape[2] // 2))
        h = self.resnet("encoder.mid_block.resnets.0", h)
        h = self._mid_attention(h, "encoder.mid_block.attentions.0")
        h = self.resnet("encoder.mid_block.resnets.1", h)
        h = ops.groupnorm(h, p["encoder.conv_norm_out.g"], p["encoder.conv_norm_out.b"], cfg["groups"], 1e-6, SILU)
        h = ops.conv(h, p["encoder.conv_out.w"], p["encoder.conv_out.b"], kh=3, kw=3, pad=1)
        return ops.conv(h, p["quant_conv.w"], p["quant_conv.b"])


class CLIPText:
    """CLIPTextModel text tower: last_hidden_state after the final LayerNorm."""

    def __init__(self, sd, cfg, dev, dtype):
        self.cfg, self.dev, self.dtype = cfg, dev, dtype
        pk = self.pk = _Packed(sd, dev, dtype)
        p = self.p = pk.p
        e = "text_model.embeddings"
        p["tok"] = sd[e + ".token_embedding.weight"].to(dev, dtype).contiguous()
        p["pos"] = sd[e + ".position_embedding.weight"].to(dev, dtype).contiguous()
        for i in range(cfg["layers"]):
            lp = f"text_model.encoder.layers.{i}"
            a = lp + ".self_attn"
            p[a + ".qk.w"] = torch.cat([sd[a + ".q_proj.weight"], sd[a + ".k_proj.weight"]], 0).contiguous().to(dev, dtype)
            p[a + ".qk.b"] = _f32(torch.cat([sd[a + ".q_proj.bias"], sd[a + ".k_proj.bias"]]), dev)
            p[a + ".v.w"] = sd[a + ".v_proj.weight"].contiguous().to(dev, dtype)
            wo = sd[a + ".out_proj.weight"]
            p[a + ".o.w"] = wo.contiguous().to(dev, dtype)
            p[a + ".o.b"] = _f32(sd[a + ".out_proj.bias"] + wo @ sd[a + ".v_proj.bias"], dev)
            pk.norm(lp + ".layer_norm1")
            pk.norm(lp + ".layer_norm2")
            pk.linear(lp + ".mlp.fc1")
            pk.linear(lp + ".mlp.fc2")
        pk.norm("text_model.final_layer_norm")
        if "text_projection.weight" in sd:      # CLIPTextModelWithProjection (SDXL text_encoder_2)
            p["text_projection.w"] = sd["text_projection.weight"].contiguous().to(dev, dtype)
        pk.sd = None

    def forward(self, ids, ctx=None, ctx_begin=2, penultimate=False):
        """ids: int [B,n] (device) -> [B,n,width]; with ctx [B,nctx,width] (BLIP-Diffusion subject tokens) the
        sequence is the prompt with ctx spliced in at `ctx_begin` (ContextCLIPTextModel), n + nctx long.
        penultimate=True is the SDXL encode_prompt form: returns (hidden_states[-2], text_embeds) -- the output of
        the second-to-last layer without the final LayerNorm, and (towers with a text_projection only) the projected
        final-LayerNorm state at the EOS position (first position of the largest id), else None."""
        cfg, p = self.cfg, self.p
        mlp_act = ops.ACT_GELU if cfg.get("act") == "gelu" else ops.ACT_QUICK_GELU
        want_pooled = penultimate and "text_projection.w" in p
        n_run = cfg["layers"] if (not penultimate or want_pooled) else cfg["layers"] - 1
        hidden2 = None
        b, n = ids.shape
        c = cfg["width"]
        if ctx is None:
            x = ops.embed_tokens(ids, p["tok"], p["pos"], n).view(b, n, c)
        else:
            x = ops.embed_tokens_ctx(ids, ctx.to(self.dtype), ctx_begin, p["tok"], p["pos"])
            n = x.shape[1]
        for i in range(n_run):
            if i == cfg["layers"] - 1:
                hidden2 = x
            lp = f"text_model.encoder.layers.{i}"
            a = lp + ".self_attn"
            h = ops.layernorm(x, p[lp + ".layer_norm1.g"], p[lp + ".layer_norm1.b"])
            qk = ops.linear(h, p[a + ".qk.w"], p[a + ".qk.b"])
            vt = project_vt(h, p[a + ".v.w"], n)
            o = attention_core(qk[:, :, :c], qk[:, :, c:], vt, cfg["heads"], n, n, causal=True)
            x = ops.linear(o, p[a + ".o.w"], p[a + ".o.b"], residual=x)
            h = ops.layernorm(x, p[lp + ".layer_norm2.g"], p[lp + ".layer_norm2.b"])
            h = ops.activation(ops.linear(h, p[lp + ".mlp.fc1.w"], p[lp + ".mlp.fc1.b"]), mlp_act)
            x = ops.linear(h, p[lp + ".mlp.fc2.w"], p[lp + ".mlp.fc2.b"], residual=x)
        if not penultimate:
            return ops.layernorm(x, p["text_model.final_layer_norm.g"], p["text_model.final_layer_norm.b"])
        if not want_pooled:
            return x, None
        eos = ids.argmax(dim=-1)                                        # index bookkeeping only
        rows = x[torch.arange(b, device=x.device), eos].contiguous()    # gather of the EOS rows (data movement)
        rows = ops.layernorm(rows, p["text_model.final_layer_norm.g"], p["text_model.final_layer_norm.b"])
        return hidden2, ops.linear(rows, p["text_projection.w"])


class SafetyChecker:
    """StableDiffusionSafetyChecker: CLIP ViT-L/14 vision tower -> post_layernorm(CLS) ->
    visual_projection -> cosine similarity against 17 concept / 3 special-care embeddings.  The tower, projection and
    the (image x concept) similarity GEMM run in the HIP kernels; the thresholding of the 20 scores per image
    (round(., 3), the 0.01 special-care adjustment) is host control flow exactly as upstream.
    Patch embedding = one implicit-GEMM conv (14x14 window, stride 14) over the normalised channels-last pixels."""

    def __init__(self, sd, cfg, dev, dtype):
        self.cfg, self.dev, self.dtype = cfg, dev, dtype
        pk = self.pk = _Packed(sd, dev, dtype)
        p = self.p = pk.p
        v = "vision_model.vision_model"
        w = cfg["width"]
        pk.conv(v + ".embeddings.patch_embedding")
        pos = sd[v + ".embeddings.position_embedding.weight"]
        p["cls_pos"] = (sd[v + ".embeddings.class_embedding"] + pos[0]).to(dev, dtype).reshape(1, 1, w).contiguous()
        p["patch_pos"] = pos[1:].to(dev, dtype).contiguous()
        pk.norm(v + ".pre_layrnorm")
        for i in range(cfg["layers"]):
            lp = f"{v}.encoder.layers.{i}"
            a = lp + ".self_attn"
            p[a + ".qk.w"] = torch.cat([sd[a + ".q_proj.weight"], sd[a + ".k_proj.weight"]], 0).contiguous().to(dev, dtype)
            p[a + ".qk.b"] = _f32(torch.cat([sd[a + ".q_proj.bias"], sd[a + ".k_proj.bias"]]), dev)
            p[a + ".v.w"] = sd[a + ".v_proj.weight"].contiguous().to(dev, dtype)
            wo = sd[a + ".out_proj.weight"]
            p[a + ".o.w"] = wo.contiguous().to(dev, dtype)
            p[a + ".o.b"] = _f32(sd[a + ".out_proj.bias"] + wo @ sd[a + ".v_proj.bias"], dev)
            pk.norm(lp + ".layer_norm1")
            pk.norm(lp + ".layer_norm2")
            pk.linear(lp + ".mlp.fc1")
            pk.linear(lp + ".mlp.fc2")
        pk.norm(v + ".post_layernorm")
        # projection + similarity stay fp32 (20 thresholds decide on the third decimal)
        p["proj.w"] = sd["visual_projection.weight"].contiguous().to(dev, torch.float32)
        emb = torch.cat([sd["special_care_embeds"], sd["concept_embeds"]], 0)
        p["embeds_n"] = torch.nn.functional.normalize(emb.float()).contiguous().to(dev)        # constant: unit rows
        self.special_w = sd["special_care_embeds_weights"].double().numpy()
        self.concept_w = sd["concept_embeds_weights"].double().numpy()
        p["special_w"] = sd["special_care_embeds_weights"].double().contiguous().to(dev)
        p["concept_w"] = sd["concept_embeds_weights"].double().contiguous().to(dev)
        # upstream decides on `round(score, 3) > 0`; round() is monotone, so that is `score >= t` for the smallest
        # double t that rounds to a positive value -- found once by walking down from 0.0005
        t = 0.0005
        while round(t, 3) > 0:
            t = math.nextafter(t, 0.0)
        self.round_threshold = math.nextafter(t, 1.0)
        pk.sd = None

    def image_embeds(self, pixels):
        """[B,S,S,8] normalised channels-last pixels -> fp32 [B, proj_dim]."""
        cfg, p = self.cfg, self.p
        v = "vision_model.vision_model"
        b = pixels.shape[0]
        ps, w, heads = cfg["patch"], cfg["width"], cfg["heads"]
        g = pixels.shape[1] // ps
        pos = p["patch_pos"].view(1, g, g, w).expand(b, -1, -1, -1).contiguous()
        # non-overlapping patches: view the image as B*g strips of ps rows whose "pixels" are whole patch rows
        # (ps*8 channels), so the patch embedding is a ps x 1 window conv (ps taps) with the SAME packed weight
        # (K index (ky*ps + kx)*8 + c either way) -- the 14 x 14 = 196-tap form exceeds the kernel's tap table
        strips = pixels.contiguous().view(b * g, ps, g, ps * 8)
        tok = ops.conv(strips, p[v + ".embeddings.patch_embedding.w"], None, kh=ps, kw=1, stride=1, pad=0,
                       residual=pos.view(b * g, 1, g, w)).view(b, g, g, -1)
        x = torch.cat([p["cls_pos"].expand(b, -1, -1), tok.view(b, g * g, w)], 1).contiguous()
        n = x.shape[1]
        x = ops.layernorm(x, p[v + ".pre_layrnorm.g"], p[v + ".pre_layrnorm.b"])
        for i in range(cfg["layers"]):
            lp = f"{v}.encoder.layers.{i}"
            a = lp + ".self_attn"
            h = ops.layernorm(x, p[lp + ".layer_norm1.g"], p[lp + ".layer_norm1.b"])
            qk = ops.linear(h, p[a + ".qk.w"], p[a + ".qk.b"])
            vt = project_vt(h, p[a + ".v.w"], n)
            o = attention_core(qk[:, :, :w], qk[:, :, w:], vt, heads, n, n)
            x = ops.linear(o, p[a + ".o.w"], p[a + ".o.b"], residual=x)
            h = ops.layernorm(x, p[lp + ".layer_norm2.g"], p[lp + ".layer_norm2.b"])
            h = ops.activation(ops.linear(h, p[lp + ".mlp.fc1.w"], p[lp + ".mlp.fc1.b"]), ops.ACT_QUICK_GELU)
            x = ops.linear(h, p[lp + ".mlp.fc2.w"], p[lp + ".mlp.fc2.b"], residual=x)
        cls = x[:, 0].contiguous()
        pooled = ops.layernorm(cls, p[v + ".post_layernorm.g"], p[v + ".post_layernorm.b"]).float()
        return ops.linear(pooled, p["proj.w"])

    def similarity(self, pixels):
        """-> (dots fp32 [B, n_special + n_concepts] = e . unit(embed_j), special-care columns first; gram fp32 [B, >=B]
        = e e^T whose diagonal is |e|^2).  Two tiny GEMMs; the division by |e| happens with the thresholding on the host."""
        e = self.image_embeds(pixels)[:, :self.cfg["proj_dim"]].contiguous()
        return ops.linear(e, self.p["embeds_n"]), ops.linear(e, e)

    def decide(self, dots, gram):
        """Upstream's per-image thresholding on the HOST (diagnostics / tests; `forward` takes the same decision on the
        device): cos = dots / |e|, scores rounded to 3 decimals, +0.01 once a special-care concept fires ->
        (flags, concept scores, special scores)."""
        dots = dots.double().cpu().numpy()
        norm = np.sqrt(np.diagonal(gram.double().cpu().numpy()[:, :dots.shape[0]]))
        ns, ncp = len(self.special_w), len(self.concept_w)
        cos = dots[:, :ns + ncp] / norm[:, None]
        flags, cs, ss = [], [], []
        for i in range(cos.shape[0]):
            adj = 0.0
            s_scores = []
            for j in range(ns):                        # upstream: the adjustment switches on inside this loop
                s_scores.append(round(float(cos[i, j] - self.special_w[j] + adj), 3))
                if s_scores[-1] > 0:
                    adj = 0.01
            c_scores = [round(float(cos[i, ns + j] - self.concept_w[j] + adj), 3) for j in range(ncp)]
            flags.append(any(v > 0 for v in c_scores))
            cs.append(c_scores)
            ss.append(s_scores)
        return flags, cs, ss

    @torch.no_grad()
    def forward(self, images_u8):
        """device u8 [B,H,W,3] decoded images -> (the same tensor with flagged images blacked out IN PLACE, device int32
        flags [B]).  No host round trip: the decision of `decide` runs in `saspa_safety_decide` (fp64, same order)."""
        from . import _lib
        from .imageproc import clip_image_preprocess
        import ctypes as C
        if not images_u8.is_contiguous():
            images_u8 = images_u8.contiguous()
        b = images_u8.shape[0]
        px = clip_image_preprocess(images_u8, self.dtype, self.cfg["image_size"])
        dots, gram = self.similarity(px)
        flags = torch.empty((b,), device=images_u8.device, dtype=torch.int32)
        lib = _lib.load()
        _lib.check(lib.saspa_safety_decide(
            C.c_void_p(dots.data_ptr()), dots.stride(0), C.c_void_p(gram.data_ptr()), gram.stride(0), b,
            C.c_void_p(self.p["special_w"].data_ptr()), len(self.special_w), C.c_void_p(self.p["concept_w"].data_ptr()),
            len(self.concept_w), C.c_double(self.round_threshold), C.c_void_p(images_u8.data_ptr()),
            images_u8[0].numel(), C.c_void_p(flags.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
            "saspa_safety_decide")
        return images_u8, flags
