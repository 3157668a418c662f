"""ORACLE (test infrastructure, NOT product code).

CPU restatement, in plain PyTorch ops, of the SDXL-base UNet that the reference
calls at train.py:2760-2761 (`unet(noisy_latents, timesteps, embeds,
added_cond_kwargs={"text_embeds", "time_ids"}).sample`).  The arithmetic itself
lives in the third-party `diffusers` package (requirements.txt:2,
`diffusers>=0.32.0`, a floor, not vendored, not installed here), so this file
restates diffusers' published UNet2DConditionModel algorithm for the SDXL-base
config and is anchored on the reference's own structural pins:

  * the SD<->diffusers key map, train.py:2418-2447 (3 down blocks x 2 resnets,
    attentions only for i>0; 3 up blocks x 3 resnets, attentions for i<2;
    mid = resnet, attn, resnet; time_embed / label_emb / conv_in / out.0 / out.2)
  * the published SDXL-base size: 2,567,463,684 parameters in 1680 tensors
    (checked in tests/test_oracle_golden.py).

PARITY UNPINNED at the diffusers boundary: the reference holds no golden vectors
for the UNet forward, and diffusers cannot be imported here.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Every parameter keeps its diffusers name so that the reference's
freeze keywords (train.py:2664-2667) and key map apply unchanged.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F


@dataclass
class UNetConfig:
    """SDXL-base values are the defaults."""
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280)
    transformer_layers: Tuple[int, ...] = (0, 2, 10)   # per level; 0 = no attention
    layers_per_block: int = 2
    head_dim: int = 64
    cross_attention_dim: int = 2048
    addition_time_embed_dim: int = 256
    pooled_dim: int = 1280
    norm_groups: int = 32
    time_embed_dim: int = field(default=0)  # 0 -> 4 * block_out_channels[0]

    def __post_init__(self):
        if self.time_embed_dim == 0:
            self.time_embed_dim = 4 * self.block_out_channels[0]

    @property
    def add_in_dim(self) -> int:
        return self.pooled_dim + 6 * self.addition_time_embed_dim


SDXL_BASE = UNetConfig()


def mini_config(c0: int = 32, layers=(0, 1, 2), ctx_dim: int = 64, pooled: int = 32,
                add_dim: int = 16, head_dim: int = 16, groups: int = 8) -> UNetConfig:
    """Same topology as SDXL-base with small widths, for CPU-sized parity tests."""
    return UNetConfig(block_out_channels=(c0, 2 * c0, 4 * c0), transformer_layers=tuple(layers),
                      head_dim=head_dim, cross_attention_dim=ctx_dim, addition_time_embed_dim=add_dim,
                      pooled_dim=pooled, norm_groups=groups)


# --------------------------------------------------------------------------------------
# parameter table: diffusers names, shapes, in diffusers' named_parameters() order
# --------------------------------------------------------------------------------------

def _resnet_params(prefix: str, cin: int, cout: int, temb: int) -> List[Tuple[str, Tuple[int, ...]]]:
    p = [
        (f"{prefix}.norm1.weight", (cin,)), (f"{prefix}.norm1.bias", (cin,)),
        (f"{prefix}.conv1.weight", (cout, cin, 3, 3)), (f"{prefix}.conv1.bias", (cout,)),
        (f"{prefix}.time_emb_proj.weight", (cout, temb)), (f"{prefix}.time_emb_proj.bias", (cout,)),
        (f"{prefix}.norm2.weight", (cout,)), (f"{prefix}.norm2.bias", (cout,)),
        (f"{prefix}.conv2.weight", (cout, cout, 3, 3)), (f"{prefix}.conv2.bias", (cout,)),
    ]
    if cin != cout:
        p += [(f"{prefix}.conv_shortcut.weight", (cout, cin, 1, 1)), (f"{prefix}.conv_shortcut.bias", (cout,))]
    return p


def _transformer_params(prefix: str, c: int, n_layers: int, ctx: int) -> List[Tuple[str, Tuple[int, ...]]]:
    p = [(f"{prefix}.norm.weight", (c,)), (f"{prefix}.norm.bias", (c,)),
         (f"{prefix}.proj_in.weight", (c, c)), (f"{prefix}.proj_in.bias", (c,))]
    for i in range(n_layers):
        b = f"{prefix}.transformer_blocks.{i}"
        p += [
            (f"{b}.norm1.weight", (c,)), (f"{b}.norm1.bias", (c,)),
            (f"{b}.attn1.to_q.weight", (c, c)), (f"{b}.attn1.to_k.weight", (c, c)),
            (f"{b}.attn1.to_v.weight", (c, c)),
            (f"{b}.attn1.to_out.0.weight", (c, c)), (f"{b}.attn1.to_out.0.bias", (c,)),
            (f"{b}.norm2.weight", (c,)), (f"{b}.norm2.bias", (c,)),
            (f"{b}.attn2.to_q.weight", (c, c)), (f"{b}.attn2.to_k.weight", (c, ctx)),
            (f"{b}.attn2.to_v.weight", (c, ctx)),
            (f"{b}.attn2.to_out.0.weight", (c, c)), (f"{b}.attn2.to_out.0.bias", (c,)),
            (f"{b}.norm3.weight", (c,)), (f"{b}.norm3.bias", (c,)),
            (f"{b}.ff.net.0.proj.weight", (8 * c, c)), (f"{b}.ff.net.0.proj.bias", (8 * c,)),
            (f"{b}.ff.net.2.weight", (c, 4 * c)), (f"{b}.ff.net.2.bias", (c,)),
        ]
    p += [(f"{prefix}.proj_out.weight", (c, c)), (f"{prefix}.proj_out.bias", (c,))]
    return p


def up_block_resnet_channels(cfg: UNetConfig):
    """(cin, cout) of every up-block resnet, following the skip stack
    (cat([hidden, skip], dim=1); skip channels popped in reverse push order)."""
    ch = cfg.block_out_channels
    nlev = len(ch)
    skips = [ch[0]]
    for i in range(nlev):
        skips += [ch[i]] * cfg.layers_per_block
        if i < nlev - 1:
            skips.append(ch[i])
    out = []
    prev = ch[-1]
    for i in range(nlev):
        cout = ch[nlev - 1 - i]
        blk = []
        for _ in range(cfg.layers_per_block + 1):
            s = skips.pop()
            blk.append((prev + s, cout))
            prev = cout
        out.append(blk)
    return out


def param_table(cfg: UNetConfig = SDXL_BASE) -> List[Tuple[str, Tuple[int, ...]]]:
    ch = cfg.block_out_channels
    nlev = len(ch)
    T = cfg.time_embed_dim
    p: List[Tuple[str, Tuple[int, ...]]] = []
    p += [("conv_in.weight", (ch[0], cfg.in_channels, 3, 3)), ("conv_in.bias", (ch[0],))]
    p += [("time_embedding.linear_1.weight", (T, ch[0])), ("time_embedding.linear_1.bias", (T,)),
          ("time_embedding.linear_2.weight", (T, T)), ("time_embedding.linear_2.bias", (T,))]
    p += [("add_embedding.linear_1.weight", (T, cfg.add_in_dim)), ("add_embedding.linear_1.bias", (T,)),
          ("add_embedding.linear_2.weight", (T, T)), ("add_embedding.linear_2.bias", (T,))]
    # down blocks (cross-attn blocks register `attentions` before `resnets`)
    prev = ch[0]
    for i in range(nlev):
        cout = ch[i]
        pre = f"down_blocks.{i}"
        nl = cfg.transformer_layers[i]
        if nl > 0:
            for j in range(cfg.layers_per_block):
                p += _transformer_params(f"{pre}.attentions.{j}", cout, nl, cfg.cross_attention_dim)
        for j in range(cfg.layers_per_block):
            p += _resnet_params(f"{pre}.resnets.{j}", prev if j == 0 else cout, cout, T)
        if i < nlev - 1:
            p += [(f"{pre}.downsamplers.0.conv.weight", (cout, cout, 3, 3)),
                  (f"{pre}.downsamplers.0.conv.bias", (cout,))]
        prev = cout
    # up blocks
    upch = up_block_resnet_channels(cfg)
    for i in range(nlev):
        lev = nlev - 1 - i
        cout = ch[lev]
        pre = f"up_blocks.{i}"
        nl = cfg.transformer_layers[lev]
        if nl > 0:
            for j in range(cfg.layers_per_block + 1):
                p += _transformer_params(f"{pre}.attentions.{j}", cout, nl, cfg.cross_attention_dim)
        for j in range(cfg.layers_per_block + 1):
            cin, co = upch[i][j]
            p += _resnet_params(f"{pre}.resnets.{j}", cin, co, T)
        if i < nlev - 1:
            p += [(f"{pre}.upsamplers.0.conv.weight", (cout, cout, 3, 3)),
                  (f"{pre}.upsamplers.0.conv.bias", (cout,))]
    # mid block
    cm = ch[-1]
    p += _transformer_params("mid_block.attentions.0", cm, cfg.transformer_layers[-1], cfg.cross_attention_dim)
    p += _resnet_params("mid_block.resnets.0", cm, cm, T)
    p += _resnet_params("mid_block.resnets.1", cm, cm, T)
    p += [("conv_norm_out.weight", (ch[0],)), ("conv_norm_out.bias", (ch[0],)),
          ("conv_out.weight", (cfg.out_channels, ch[0], 3, 3)), ("conv_out.bias", (cfg.out_channels,))]
    return p


def init_params(cfg: UNetConfig = SDXL_BASE, seed: int = 1234, dtype=torch.float32,
                scale_out: float = 1.0) -> Dict[str, torch.Tensor]:
    """Seed-generated weights (no SDXL checkpoint is reachable: no network).
    PyTorch default layer init (kaiming-uniform a=sqrt(5) => U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    for Linear/Conv weight and bias; norm weight 1, bias 0), drawn in param_table order from
    a CPU generator with the given seed."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    table = param_table(cfg)
    shapes = dict(table)
    for name, shape in table:
        is_norm = (".norm" in name or name.startswith("conv_norm_out"))
        if is_norm:
            t = torch.ones(shape) if name.endswith(".weight") else torch.zeros(shape)
        else:
            wname = name[:-5] + ".weight" if name.endswith(".bias") else name
            wshape = shapes[wname]
            fan_in = 1
            for d in wshape[1:]:
                fan_in *= d
            bound = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shape, generator=g) * 2.0 - 1.0) * bound
        out[name] = t.to(dtype)
    return out


# --------------------------------------------------------------------------------------
# forward (functional; autograd supplies the backward)
# --------------------------------------------------------------------------------------

def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers get_timestep_embedding with flip_sin_to_cos=True, downscale_freq_shift=0,
    scale=1, max_period=10000 -> [cos | sin], fp32."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    freqs = torch.exp(exponent)
    args = t.reshape(-1).float()[:, None] * freqs[None, :]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class RefUNet:
    """Functional UNet over a {name: tensor} dict. `autocast_bf16=True` reproduces the
    reference dataflow (bf16 params, torch.autocast(bf16): matmul/conv in bf16, norms /
    softmax in fp32); False = pure fp32 (the 1e-3 parity oracle)."""

    def __init__(self, cfg: UNetConfig, params: Dict[str, torch.Tensor]):
        self.cfg = cfg
        self.p = params

    # -- leaf ops ---------------------------------------------------------------------
    def _lin(self, x, name, bias=True):
        return F.linear(x, self.p[name + ".weight"], self.p[name + ".bias"] if bias else None)

    def _conv(self, x, name, stride=1, pad=1):
        return F.conv2d(x, self.p[name + ".weight"], self.p[name + ".bias"], stride=stride, padding=pad)

    def _gn(self, x, name, eps):
        return F.group_norm(x, self.cfg.norm_groups, self.p[name + ".weight"], self.p[name + ".bias"], eps)

    def _ln(self, x, name):
        return F.layer_norm(x, (x.shape[-1],), self.p[name + ".weight"], self.p[name + ".bias"], 1e-5)

    # -- blocks -----------------------------------------------------------------------
    def resnet(self, x, temb, pre):
        h = self._conv(F.silu(self._gn(x, pre + ".norm1", 1e-5)), pre + ".conv1")
        t = self._lin(F.silu(temb), pre + ".time_emb_proj")
        h = h + t[:, :, None, None].to(h.dtype)
        h = self._conv(F.silu(self._gn(h, pre + ".norm2", 1e-5)), pre + ".conv2")
        if (pre + ".conv_shortcut.weight") in self.p:
            x = self._conv(x, pre + ".conv_shortcut", pad=0)
        return x + h

    def attention(self, x, ctx, pre):
        B, T, C = x.shape
        hd = self.cfg.head_dim
        nh = C // hd
        q = self._lin(x, pre + ".to_q", bias=False)
        k = self._lin(ctx, pre + ".to_k", bias=False)
        v = self._lin(ctx, pre + ".to_v", bias=False)
        q = q.view(B, T, nh, hd).transpose(1, 2)
        k = k.view(B, -1, nh, hd).transpose(1, 2)
        v = v.view(B, -1, nh, hd).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False)  # scale 1/sqrt(hd)
        o = o.transpose(1, 2).reshape(B, T, C)
        return self._lin(o, pre + ".to_out.0")

    def tblock(self, h, ctx, pre):
        n = self._ln(h, pre + ".norm1")
        h = self.attention(n, n, pre + ".attn1") + h
        n = self._ln(h, pre + ".norm2")
        h = self.attention(n, ctx, pre + ".attn2") + h
        n = self._ln(h, pre + ".norm3")
        proj = self._lin(n, pre + ".ff.net.0.proj")
        val, gate = proj.chunk(2, dim=-1)
        h = self._lin(val * F.gelu(gate), pre + ".ff.net.2") + h
        return h

    def transformer(self, x, ctx, pre, n_layers):
        B, C, H, W = x.shape
        res = x
        h = self._gn(x, pre + ".norm", 1e-6)
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = self._lin(h, pre + ".proj_in")
        for i in range(n_layers):
            h = self.tblock(h, ctx, f"{pre}.transformer_blocks.{i}")
        h = self._lin(h, pre + ".proj_out")
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        return h + res

    # -- whole model --------------------------------------------------------------------
    def embed(self, timesteps, pooled, time_ids, dtype):
        cfg = self.cfg
        t_emb = timestep_embedding(timesteps, cfg.block_out_channels[0]).to(dtype)
        emb = self._lin(F.silu(self._lin(t_emb, "time_embedding.linear_1")), "time_embedding.linear_2")
        tid = timestep_embedding(time_ids.flatten(), cfg.addition_time_embed_dim)
        tid = tid.reshape(time_ids.shape[0], -1)
        add = torch.cat([pooled, tid.to(pooled.dtype)], dim=-1).to(emb.dtype)
        aug = self._lin(F.silu(self._lin(add, "add_embedding.linear_1")), "add_embedding.linear_2")
        return emb + aug

    def forward(self, sample, timesteps, ctx, pooled, time_ids, autocast_bf16=False):
        if autocast_bf16:
            with torch.autocast(device_type="cpu", dtype=torch.bfloat16):
                return self._forward(sample, timesteps, ctx, pooled, time_ids)
        return self._forward(sample, timesteps, ctx, pooled, time_ids)

    def _forward(self, sample, timesteps, ctx, pooled, time_ids):
        cfg = self.cfg
        ch = cfg.block_out_channels
        nlev = len(ch)
        emb = self.embed(timesteps, pooled, time_ids, sample.dtype)
        h = self._conv(sample, "conv_in")
        skips = [h]
        for i in range(nlev):
            pre = f"down_blocks.{i}"
            for j in range(cfg.layers_per_block):
                h = self.resnet(h, emb, f"{pre}.resnets.{j}")
                if cfg.transformer_layers[i] > 0:
                    h = self.transformer(h, ctx, f"{pre}.attentions.{j}", cfg.transformer_layers[i])
                skips.append(h)
            if i < nlev - 1:
                h = self._conv(h, f"{pre}.downsamplers.0.conv", stride=2, pad=1)
                skips.append(h)
        h = self.resnet(h, emb, "mid_block.resnets.0")
        h = self.transformer(h, ctx, "mid_block.attentions.0", cfg.transformer_layers[-1])
        h = self.resnet(h, emb, "mid_block.resnets.1")
        for i in range(nlev):
            lev = nlev - 1 - i
            pre = f"up_blocks.{i}"
            for j in range(cfg.layers_per_block + 1):
                h = torch.cat([h, skips.pop()], dim=1)
                h = self.resnet(h, emb, f"{pre}.resnets.{j}")
                if cfg.transformer_layers[lev] > 0:
                    h = self.transformer(h, ctx, f"{pre}.attentions.{j}", cfg.transformer_layers[lev])
            if i < nlev - 1:
                h = F.interpolate(h, scale_factor=2.0, mode="nearest")
                h = self._conv(h, f"{pre}.upsamplers.0.conv")
        h = F.silu(self._gn(h, "conv_norm_out", 1e-5))
        return self._conv(h, "conv_out")


def forward_macs(cfg: UNetConfig, h: int, w: int, ctx_len: int = 77) -> int:
    """Forward multiply-accumulates per sample (conv / linear / attention matmuls only),
    the figure BASELINE.md section 3 prices the roofline from (3.381 TMAC @128x128)."""
    ch = cfg.block_out_channels
    nlev = len(ch)
    T = cfg.time_embed_dim
    total = 0

    def conv(cin, cout, hh, ww, k=3):
        return cin * cout * k * k * hh * ww

    def resnet(cin, cout, hh, ww):
        m = conv(cin, cout, hh, ww) + conv(cout, cout, hh, ww) + T * cout
        if cin != cout:
            m += conv(cin, cout, hh, ww, 1)
        return m

    def transformer(c, nl, hh, ww):
        t = hh * ww
        m = 2 * t * c * c  # proj_in/out
        per = 4 * t * c * c                   # q,k,v,out self
        per += 2 * t * t * c                  # QK^T + PV self
        per += 2 * t * c * c                  # q, out cross
        per += 2 * ctx_len * cfg.cross_attention_dim * c   # k,v cross
        per += 2 * t * ctx_len * c            # cross matmuls
        per += t * c * 8 * c + t * 4 * c * c  # GEGLU ff
        return m + nl * per

    total += ch[0] * T + T * T + cfg.add_in_dim * T + T * T
    total += conv(cfg.in_channels, ch[0], h, w)
    hh, ww = h, w
    prev = ch[0]
    for i in range(nlev):
        for j in range(cfg.layers_per_block):
            total += resnet(prev if j == 0 else ch[i], ch[i], hh, ww)
            if cfg.transformer_layers[i] > 0:
                total += transformer(ch[i], cfg.transformer_layers[i], hh, ww)
        prev = ch[i]
        if i < nlev - 1:
            hh, ww = hh // 2, ww // 2
            total += conv(ch[i], ch[i], hh, ww)
    total += 2 * resnet(ch[-1], ch[-1], hh, ww) + transformer(ch[-1], cfg.transformer_layers[-1], hh, ww)
    upch = up_block_resnet_channels(cfg)
    for i in range(nlev):
        lev = nlev - 1 - i
        for j in range(cfg.layers_per_block + 1):
            cin, cout = upch[i][j]
            total += resnet(cin, cout, hh, ww)
            if cfg.transformer_layers[lev] > 0:
                total += transformer(cout, cfg.transformer_layers[lev], hh, ww)
        if i < nlev - 1:
            hh, ww = hh * 2, ww * 2
            total += conv(ch[lev], ch[lev], hh, ww)
    total += conv(ch[0], cfg.out_channels, hh, ww)
    return total
