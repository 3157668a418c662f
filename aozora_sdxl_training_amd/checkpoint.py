"""Checkpoint / resume seams of the training step.

What the reference does around the step and what is mirrored here (host-side logic only; no kernels):

  * single-file SDXL <-> diffusers parameter names        train.py:2418-2465  -> unet_key_mapping()
  * load the UNet out of a single-file checkpoint          train.py:1437-1469  -> read_unet_state() / load_unet()
    (the reference goes through diffusers' from_single_file; here the inverse key map feeds
     AozoraUNet.load_state_dict directly)
  * merge the trained UNet back into the base checkpoint   train.py:2467-2511  -> save_model()
  * training-state file (.pt) next to the model            train.py:2513-2531  -> save_training_state()
  * resume                                                 train.py:2558-2573, 2684-2688 -> load_training_state(), restore_rng()
  * emergency-save flag file                               train.py:2534-2542  -> consume_force_save_flag()

The key map is derived here from the SDXL block structure (3 down blocks x 2 resnets, attentions for i > 0;
3 up blocks x 3 resnets, attentions for i < 2; mid = resnet, attention, resnet) instead of the reference's chain
of string replacements; tests/test_checkpoint.py pins all 1680 names against a digest of the reference's own
mapping (tests/golden/golden_host.json "keymap").
"""
from __future__ import annotations

import os
import random
import re
from pathlib import Path
from typing import Dict, Iterable, Mapping, Optional

import numpy as np
import torch

PREFIX = "model.diffusion_model."

_STATIC = {
    "time_embedding.linear_1": "time_embed.0", "time_embedding.linear_2": "time_embed.2",
    "add_embedding.linear_1": "label_emb.0.0", "add_embedding.linear_2": "label_emb.0.2",
    "conv_in": "input_blocks.0.0", "conv_norm_out": "out.0", "conv_out": "out.2",
}
_RESNET = {"norm1": "in_layers.0", "conv1": "in_layers.2", "norm2": "out_layers.0", "conv2": "out_layers.3",
           "time_emb_proj": "emb_layers.1", "conv_shortcut": "skip_connection"}

_RE_DOWN = re.compile(r"^down_blocks\.(\d+)\.(resnets|attentions)\.(\d+)\.(.*)$")
_RE_DOWNS = re.compile(r"^down_blocks\.(\d+)\.downsamplers\.0\.conv\.(.*)$")
_RE_UP = re.compile(r"^up_blocks\.(\d+)\.(resnets|attentions)\.(\d+)\.(.*)$")
_RE_UPS = re.compile(r"^up_blocks\.(\d+)\.upsamplers\.0\.(.*)$")
_RE_MID = re.compile(r"^mid_block\.(resnets|attentions)\.(\d+)\.(.*)$")


def _resnet_tail(rest: str) -> str:
    head, _, leaf = rest.partition(".")
    return f"{_RESNET[head]}.{leaf}" if head in _RESNET else rest


def _sd_name(name: str) -> str:
    stem, _, leaf = name.rpartition(".")
    if stem in _STATIC:
        return f"{_STATIC[stem]}.{leaf}"
    m = _RE_DOWN.match(name)
    if m:
        i, kind, j, rest = int(m.group(1)), m.group(2), int(m.group(3)), m.group(4)
        blk = 3 * i + j + 1
        return f"input_blocks.{blk}.0.{_resnet_tail(rest)}" if kind == "resnets" else f"input_blocks.{blk}.1.{rest}"
    m = _RE_DOWNS.match(name)
    if m:
        return f"input_blocks.{3 * (int(m.group(1)) + 1)}.0.op.{m.group(2)}"
    m = _RE_UP.match(name)
    if m:
        i, kind, j, rest = int(m.group(1)), m.group(2), int(m.group(3)), m.group(4)
        blk = 3 * i + j
        return f"output_blocks.{blk}.0.{_resnet_tail(rest)}" if kind == "resnets" else f"output_blocks.{blk}.1.{rest}"
    m = _RE_UPS.match(name)
    if m:       # the upsampler follows the block's last resnet (and its attention: up blocks 0 and 1 both have one)
        return f"output_blocks.{3 * int(m.group(1)) + 2}.2.{m.group(2)}"
    m = _RE_MID.match(name)
    if m:
        kind, j, rest = m.group(1), int(m.group(2)), m.group(3)
        return f"middle_block.{2 * j}.{_resnet_tail(rest)}" if kind == "resnets" else f"middle_block.1.{rest}"
    return name


def unet_key_mapping(names: Iterable[str]) -> Dict[str, str]:
    """diffusers parameter name -> key of the single-file SDXL checkpoint (train.py:2449-2465)."""
    return {n: PREFIX + _sd_name(n) for n in names}


def read_unet_state(path, names: Iterable[str], dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    """Read the UNet out of a single-file .safetensors checkpoint as {diffusers name: tensor} (train.py:1437-1469)."""
    from safetensors import safe_open
    km = unet_key_mapping(names)
    out, missing = {}, []
    with safe_open(str(path), framework="pt", device="cpu") as f:
        keys = set(f.keys())
        for n, k in km.items():
            if k not in keys:
                missing.append(k)
                continue
            out[n] = f.get_tensor(k).to(dtype)
    if missing:
        raise KeyError(f"{len(missing)} UNet keys missing from {Path(path).name}, e.g. {missing[:3]}")
    return out


def peek_channels(path):
    """(in_channels, out_channels) probed from the checkpoint like train.py:1439-1455 (defaults 4, 4)."""
    from safetensors import safe_open
    cin = cout = 4
    with safe_open(str(path), framework="pt", device="cpu") as f:
        keys = set(f.keys())
        if PREFIX + "input_blocks.0.0.weight" in keys:
            cin = f.get_slice(PREFIX + "input_blocks.0.0.weight").get_shape()[1]
        if PREFIX + "out.2.weight" in keys:
            cout = f.get_slice(PREFIX + "out.2.weight").get_shape()[0]
    return cin, cout


def load_unet(path, device="cuda:0", cfg=None):
    """load_unet_robust (train.py:1437) for the HIP executor: AozoraUNet with the checkpoint's weights."""
    from .unet import AozoraUNet
    from .unet_spec import SDXL_BASE
    cfg = cfg if cfg is not None else SDXL_BASE
    cin, cout = peek_channels(path)
    if (cin, cout) != (cfg.in_channels, cfg.out_channels):
        raise ValueError(f"checkpoint has in/out channels {cin}/{cout}; this build runs {cfg.in_channels}/{cfg.out_channels}")
    unet = AozoraUNet(cfg, device)
    unet.load_state_dict(read_unet_state(path, [n for n, _ in unet.named_parameters()]))
    return unet


def _state_of(unet_or_state) -> Mapping[str, torch.Tensor]:
    return unet_or_state.state_dict() if hasattr(unet_or_state, "state_dict") else unet_or_state


def save_model(output_path, unet_or_state, base_checkpoint_path, compute_dtype=torch.bfloat16):
    """train.py:2467-2511: load the base single-file checkpoint, cast its floating tensors to compute_dtype, overwrite
    the UNet keys with the trained parameters, write a .safetensors.  Returns (n_merged, missing_target_keys)."""
    from safetensors.torch import load_file, save_file
    output_path = Path(output_path)
    output_path.parent.mkdir(parents=True, exist_ok=True)
    base = load_file(str(base_checkpoint_path), device="cpu")
    state = _state_of(unet_or_state)
    for k, t in base.items():
        if t.dtype in (torch.float32, torch.float16, torch.bfloat16):
            base[k] = t.to(dtype=compute_dtype)
    missing, merged = [], 0
    for hf_key, target in unet_key_mapping(list(state.keys())).items():
        if target not in base:
            missing.append(target)
        base[target] = state[hf_key].detach().to("cpu", dtype=compute_dtype).contiguous()
        merged += 1
    save_file(base, str(output_path))
    return merged, missing


def save_training_state(path, global_step, micro_step, optimizer, sampler_seed, sampler_epoch, timestep_sampler=None):
    """train.py:2513-2531 (same dict keys, so either trainer can resume the other's file)."""
    optim_state = optimizer.save_cpu_state() if hasattr(optimizer, "save_cpu_state") else optimizer.state_dict()
    st = {
        "global_step": global_step, "micro_step": micro_step, "optimizer_state": optim_state,
        "sampler_seed": sampler_seed, "sampler_epoch": max(int(sampler_epoch) - 1, 0),
        "timestep_sampler_state": timestep_sampler.state_dict() if timestep_sampler is not None and hasattr(timestep_sampler, "state_dict") else None,
        "random_state": random.getstate(), "numpy_state": np.random.get_state(),
        "torch_cpu_state": torch.get_rng_state(),
        "torch_cuda_state": torch.cuda.get_rng_state() if torch.cuda.is_available() else None,
    }
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    torch.save(st, str(path))
    return st


def load_training_state(path, grad_accum: int):
    """train.py:2558-2573: -> dict(global_step, micro_step, optimizer_step, sampler_seed, sampler_epoch,
    timestep_sampler_state, optimizer_state, raw) with the reference's defaults for absent keys."""
    st = torch.load(str(path), map_location="cpu", weights_only=False)
    gs = st.get("global_step", 0)
    micro = st.get("micro_step", gs * grad_accum)
    return dict(global_step=gs, micro_step=micro, optimizer_step=micro // grad_accum, sampler_seed=st["sampler_seed"],
                sampler_epoch=st.get("sampler_epoch", 0), timestep_sampler_state=st.get("timestep_sampler_state"),
                optimizer_state=st["optimizer_state"], raw=st)


def restore_rng(raw_state):
    """train.py:2569-2573."""
    if "random_state" in raw_state:
        random.setstate(raw_state["random_state"])
    if "numpy_state" in raw_state:
        np.random.set_state(raw_state["numpy_state"])
    if "torch_cpu_state" in raw_state:
        torch.set_rng_state(raw_state["torch_cpu_state"])
    if raw_state.get("torch_cuda_state") is not None and torch.cuda.is_available():
        torch.cuda.set_rng_state(raw_state["torch_cuda_state"])


def resume_optimizer(optimizer, optimizer_state, lr_scheduler=None, micro_step: Optional[int] = None):
    """train.py:2684-2688 (the reference swallows load errors; here they propagate)."""
    if optimizer_state:
        if hasattr(optimizer, "load_cpu_state"):
            optimizer.load_cpu_state(optimizer_state)
        else:
            optimizer.load_state_dict(optimizer_state)
    if lr_scheduler is not None and micro_step is not None:
        lr_scheduler.step(micro_step)


def consume_force_save_flag(flag_path) -> bool:
    """train.py:2534-2542: True exactly once per flag file (GUI 'emergency save' button)."""
    flag_path = Path(flag_path)
    if not flag_path.exists():
        return False
    try:
        flag_path.unlink()
        return True
    except OSError:
        return False


_ILLEGAL = re.compile(r'[<>:"/\\|?*\x00-\x1f]')


def output_model_stem(config, source_path) -> str:
    """train.py:2334-2349: stem of every file a run writes.  OUTPUT_NAME "auto" (the default) -> "<source stem>_trained_{uuid}";
    "{uuid}" -> six random [a-z0-9] characters drawn once per run; directory components and a ".safetensors" suffix are
    dropped, characters that are illegal in file names become "_".  Resolved once and cached on the config object."""
    import secrets
    import string
    cached = getattr(config, "_RESOLVED_OUTPUT_STEM", None)
    if cached:
        return cached
    src = Path(source_path).stem
    want = str(getattr(config, "OUTPUT_NAME", "auto") or "auto").strip()
    if want.lower() == "auto":
        want = src + "_trained_{uuid}"
    tag = "".join(secrets.choice(string.ascii_lowercase + string.digits) for _ in range(6))
    want = Path(want.replace("{uuid}", tag)).name
    if want.lower().endswith(".safetensors"):
        want = want[:-len(".safetensors")]
    want = _ILLEGAL.sub("_", want).strip(" .")
    stem = want if want else f"{src}_trained_{tag}"
    config._RESOLVED_OUTPUT_STEM = stem
    return stem


def checkpoint_names(output_stem: str, global_step: int):
    """File names of train.py:2515-2517."""
    return f"{output_stem}_step_{global_step}.safetensors", f"{output_stem}_training_state_step_{global_step}.pt"
