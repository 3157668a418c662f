"""Structure of the SDXL-base UNet as the product sees it: configuration, diffusers parameter
names/shapes in diffusers' named_parameters() order (train.py:2665 iterates them for the freeze
keywords; raven.py:157-168 indexes optimizer state by position in that order), and the skip-stack
channel bookkeeping.  Structural pins: train.py:2418-2447 (key map), 2,567,463,684 params / 1680
tensors.  diffusers itself is not importable here: PARITY UNPINNED."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280)
    transformer_layers: Tuple[int, ...] = (0, 2, 10)
    layers_per_block: int = 2
    head_dim: int = 64
    cross_attention_dim: int = 2048
    addition_time_embed_dim: int = 256
    pooled_dim: int = 1280
    norm_groups: int = 32
    time_embed_dim: int = 0

    def __post_init__(self):
        if self.time_embed_dim == 0:
            self.time_embed_dim = 4 * self.block_out_channels[0]

    @property
    def add_in_dim(self) -> int:
        return self.pooled_dim + 6 * self.addition_time_embed_dim


SDXL_BASE = UNetConfig()


def mini_config(c0=64, layers=(0, 1, 2), ctx_dim=128, pooled=64, add_dim=32, groups=8) -> UNetConfig:
    """SDXL topology at small width (head_dim stays 64: the attention kernel is specialised for it)."""
    return UNetConfig(block_out_channels=(c0, 2 * c0, 4 * c0), transformer_layers=tuple(layers), head_dim=64,
                      cross_attention_dim=ctx_dim, addition_time_embed_dim=add_dim, pooled_dim=pooled, norm_groups=groups)


def skip_channels(cfg: UNetConfig) -> List[int]:
    ch = cfg.block_out_channels
    s = [ch[0]]
    for i in range(len(ch)):
        s += [ch[i]] * cfg.layers_per_block
        if i < len(ch) - 1:
            s.append(ch[i])
    return s


def up_resnet_channels(cfg: UNetConfig):
    ch = cfg.block_out_channels
    skips = skip_channels(cfg)
    out, prev = [], ch[-1]
    for i in range(len(ch)):
        cout = ch[len(ch) - 1 - i]
        blk = []
        for _ in range(cfg.layers_per_block + 1):
            s = skips.pop()
            blk.append((prev, s, cout))   # (hidden channels, skip channels, out channels)
            prev = cout
        out.append(blk)
    return out


def _resnet(pre, cin, cout, T):
    p = [(f"{pre}.norm1.weight", (cin,)), (f"{pre}.norm1.bias", (cin,)),
         (f"{pre}.conv1.weight", (cout, cin, 3, 3)), (f"{pre}.conv1.bias", (cout,)),
         (f"{pre}.time_emb_proj.weight", (cout, T)), (f"{pre}.time_emb_proj.bias", (cout,)),
         (f"{pre}.norm2.weight", (cout,)), (f"{pre}.norm2.bias", (cout,)),
         (f"{pre}.conv2.weight", (cout, cout, 3, 3)), (f"{pre}.conv2.bias", (cout,))]
    if cin != cout:
        p += [(f"{pre}.conv_shortcut.weight", (cout, cin, 1, 1)), (f"{pre}.conv_shortcut.bias", (cout,))]
    return p


def _transformer(pre, c, n, ctx):
    p = [(f"{pre}.norm.weight", (c,)), (f"{pre}.norm.bias", (c,)), (f"{pre}.proj_in.weight", (c, c)), (f"{pre}.proj_in.bias", (c,))]
    for i in range(n):
        b = f"{pre}.transformer_blocks.{i}"
        p += [(f"{b}.norm1.weight", (c,)), (f"{b}.norm1.bias", (c,)),
              (f"{b}.attn1.to_q.weight", (c, c)), (f"{b}.attn1.to_k.weight", (c, c)), (f"{b}.attn1.to_v.weight", (c, c)),
              (f"{b}.attn1.to_out.0.weight", (c, c)), (f"{b}.attn1.to_out.0.bias", (c,)),
              (f"{b}.norm2.weight", (c,)), (f"{b}.norm2.bias", (c,)),
              (f"{b}.attn2.to_q.weight", (c, c)), (f"{b}.attn2.to_k.weight", (c, ctx)), (f"{b}.attn2.to_v.weight", (c, ctx)),
              (f"{b}.attn2.to_out.0.weight", (c, c)), (f"{b}.attn2.to_out.0.bias", (c,)),
              (f"{b}.norm3.weight", (c,)), (f"{b}.norm3.bias", (c,)),
              (f"{b}.ff.net.0.proj.weight", (8 * c, c)), (f"{b}.ff.net.0.proj.bias", (8 * c,)),
              (f"{b}.ff.net.2.weight", (c, 4 * c)), (f"{b}.ff.net.2.bias", (c,))]
    p += [(f"{pre}.proj_out.weight", (c, c)), (f"{pre}.proj_out.bias", (c,))]
    return p


def param_table(cfg: UNetConfig = SDXL_BASE):
    """[(diffusers name, logical shape)] in diffusers registration order: conv_in, time_embedding,
    add_embedding, down_blocks, up_blocks, mid_block, conv_norm_out, conv_out; cross-attention blocks list
    `attentions` before `resnets`."""
    ch, T, n = cfg.block_out_channels, cfg.time_embed_dim, len(cfg.block_out_channels)
    p = [("conv_in.weight", (ch[0], cfg.in_channels, 3, 3)), ("conv_in.bias", (ch[0],)),
         ("time_embedding.linear_1.weight", (T, ch[0])), ("time_embedding.linear_1.bias", (T,)),
         ("time_embedding.linear_2.weight", (T, T)), ("time_embedding.linear_2.bias", (T,)),
         ("add_embedding.linear_1.weight", (T, cfg.add_in_dim)), ("add_embedding.linear_1.bias", (T,)),
         ("add_embedding.linear_2.weight", (T, T)), ("add_embedding.linear_2.bias", (T,))]
    prev = ch[0]
    for i in range(n):
        pre, c, nl = f"down_blocks.{i}", ch[i], cfg.transformer_layers[i]
        if nl:
            for j in range(cfg.layers_per_block):
                p += _transformer(f"{pre}.attentions.{j}", c, nl, cfg.cross_attention_dim)
        for j in range(cfg.layers_per_block):
            p += _resnet(f"{pre}.resnets.{j}", prev if j == 0 else c, c, T)
        if i < n - 1:
            p += [(f"{pre}.downsamplers.0.conv.weight", (c, c, 3, 3)), (f"{pre}.downsamplers.0.conv.bias", (c,))]
        prev = c
    up = up_resnet_channels(cfg)
    for i in range(n):
        lev = n - 1 - i
        pre, c, nl = f"up_blocks.{i}", ch[lev], cfg.transformer_layers[lev]
        if nl:
            for j in range(cfg.layers_per_block + 1):
                p += _transformer(f"{pre}.attentions.{j}", c, nl, cfg.cross_attention_dim)
        for j in range(cfg.layers_per_block + 1):
            hc, sc, co = up[i][j]
            p += _resnet(f"{pre}.resnets.{j}", hc + sc, co, T)
        if i < n - 1:
            p += [(f"{pre}.upsamplers.0.conv.weight", (c, c, 3, 3)), (f"{pre}.upsamplers.0.conv.bias", (c,))]
    cm = ch[-1]
    p += _transformer("mid_block.attentions.0", cm, cfg.transformer_layers[-1], cfg.cross_attention_dim)
    p += _resnet("mid_block.resnets.0", cm, cm, T) + _resnet("mid_block.resnets.1", cm, cm, T)
    p += [("conv_norm_out.weight", (ch[0],)), ("conv_norm_out.bias", (ch[0],)),
          ("conv_out.weight", (cfg.out_channels, ch[0], 3, 3)), ("conv_out.bias", (cfg.out_channels,))]
    return p
