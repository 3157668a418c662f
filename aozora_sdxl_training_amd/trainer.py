"""The training loop around the HIP step (train.py:2544-2830 restated on this package's objects): data feed -> micro-step
(TrainStep: noise mix, UNet forward, weighted MSE, backward) -> clip -> Raven / Titan -> LR curve -> reporter ->
checkpoints / resume.  It is the caller of the hot path (SURVEY.md 8a row a1 with the 8f rows plugged in); no GUI, no
offline caching -- `config` is any object with the reference's flat attribute names (config.TrainingConfig builds one from
a GUI preset: `python -m aozora_sdxl_training_amd.trainer --config X.json`, see main()).

Deviations kept deliberately: noise / rectified-flow jitter are drawn on a CPU generator (the
reference draws on the device generator, whose stream is backend specific), and the three per-micro-step `.item()` syncs of
the reference collapse into one read of the loss scalar, taken one micro-step late so that the device queue never drains.
"""
from __future__ import annotations

import gc
import time
from collections import deque
from pathlib import Path
from typing import Optional

import torch

from . import checkpoint as ckpt
from . import data as feed
from .clip import clip_grad_norm_
from .optimizers import RavenAdamW, TitanAdamW
from .schedule import (CustomCurveLRScheduler, TimestepSampler, ddpm_alphas_cumprod, generate_noise, make_time_ids,
                       seeded_torch_generator, timestep_loss_curve_from_config, trainable_mask)
from .telemetry import Reporter
from .train_step import TrainStep

_RAVEN_DEFAULTS = dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, debias_strength=0.3, momentum_dtype="bfloat16")


def _momentum_dtype(v):
    """create_optimizer (train.py:2262-2263, 2268-2269): the string "bfloat16" selects bf16, ANY other value fp32 (fp16 state
    is only reachable by constructing RavenAdamW / TitanAdamW directly with torch.float16)."""
    if isinstance(v, torch.dtype):
        return v
    return torch.bfloat16 if v == "bfloat16" else torch.float32


class _Silent:
    def log_step(self, *a, **k): pass
    def log_message(self, *a, **k): pass
    def shutdown(self): pass


class _NoState:
    """Placeholder for the training-state file of a data-parallel run (the optimizer state lives in the per-rank shards)."""
    def save_cpu_state(self): return {"_sharded": True}


def _optimizer(config, params):
    """create_optimizer (train.py:2257-2330) for the two optimizers of the hot path."""
    kind = str(getattr(config, "OPTIMIZER_TYPE", "raven")).lower()
    curve = getattr(config, "LR_CUSTOM_CURVE", [])
    lr = max(p[1] for p in curve) if curve else config.LEARNING_RATE
    user = dict(getattr(config, "TITAN_PARAMS" if kind == "titan" else "RAVEN_PARAMS", {}) or {})
    hp = {**_RAVEN_DEFAULTS, **user}
    mdt = _momentum_dtype(hp.pop("momentum_dtype", "bfloat16"))
    cls = TitanAdamW if kind == "titan" else RavenAdamW
    return cls([{"params": params, "lr_scale": 1.0}], lr=lr, betas=tuple(hp["betas"]), eps=hp["eps"], weight_decay=hp["weight_decay"],
               debias_strength=hp["debias_strength"], momentum_dtype=mdt)


def train(config, unet=None, device="cuda:0", reporter: Optional[Reporter] = None, num_workers: Optional[int] = None):
    """Run config.MAX_TRAIN_STEPS micro-steps.  Returns dict(losses, grad_norms, lrs, micro_step, optimizer_step, saved).

    Data parallel: when torch.distributed is initialised (one process per GPU, backend nccl = RCCL), config.BATCH_SIZE
    stays the GLOBAL micro-batch: every rank builds the same schedule / tickets / noise and takes rows [r*b, (r+1)*b);
    the optimizer is dist.ShardedRaven (reduce-scatter -> clip -> sharded Raven -> all-gather, overlapped with the step);
    rank 0 reports and writes the model, every rank writes its own optimizer-state shard."""
    import torch.distributed as tdist
    dp = tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1
    world, rank = (tdist.get_world_size(), tdist.get_rank()) if dp else (1, 0)
    GA = int(config.GRADIENT_ACCUMULATION_STEPS)
    mode = getattr(config, "PREDICTION_TYPE", "epsilon")
    config.is_rectified_flow = (mode == "rectified_flow")
    micro_step = optimizer_step = 0
    model_to_load = Path(config.SINGLE_FILE_CHECKPOINT_PATH)
    sampler_seed, optimizer_state, ts_state = config.SEED, None, None
    if getattr(config, "RESUME_TRAINING", False):                                   # train.py:2558-2573
        rs = ckpt.load_training_state(config.RESUME_STATE_PATH, GA)
        micro_step, optimizer_step = rs["micro_step"], rs["optimizer_step"]
        sampler_seed, ts_state, optimizer_state = rs["sampler_seed"], rs["timestep_sampler_state"], rs["optimizer_state"]
        model_to_load = Path(config.RESUME_MODEL_PATH)
        ckpt.restore_rng(rs["raw"])
    if unet is None:
        unet = ckpt.load_unet(model_to_load, device)
    names = [n for n, _ in unet.named_parameters()]
    excl = getattr(config, "UNET_EXCLUDE_TARGETS", []) or []
    if isinstance(excl, str):                                                        # "a, b" -> ["a", "b"] (train.py:304-307)
        excl = [k.strip() for k in excl.split(",") if k.strip()]
    for (n, p), m in zip(unet.named_parameters(), trainable_mask(names, [k for k in excl if k])):
        p.requires_grad = m                                                          # train.py:2664-2667
    params = [p for p in unet.parameters() if p.requires_grad]
    # the step object first: it creates the data-gradient stream, and the order in which a process creates its streams decides
    # which hardware queues they share (streams.py); the optimizer's copy / exchange streams come after it, as in bench.py
    loss_curve = timestep_loss_curve_from_config(config, 1000)
    step = TrainStep(unet, mode=mode, grad_accum=GA, world_size=world, loss_curve=loss_curve, use_graph=False)
    from .dist import ShardedRaven, ShardedTitan
    titan = str(getattr(config, "OPTIMIZER_TYPE", "raven")).lower() == "titan"
    # Single-GPU Titan: the device-accumulator form (dist.ShardedTitan in a group of one: fp32 gradient accumulator in HBM, the same
    # arithmetic -- titan.py:119-131, 162-184, 230-296 -- tests/test_fullsize_gpu.py cfg5) unless the preset asks for the reference's
    # residency with TITAN_HOST_GRADIENTS = true (optimizers.TitanAdamW: fp32 gradients in pinned HOST memory, 10.3 GB written over the
    # host link after every micro-step).  Measured at 1024^2, B = 4 x GA 8 (bench.py other_configs, round 5): 2 991 ms per iteration
    # with host gradients against 957 ms -- the host buffer exists for 12 GB cards, this one has 288 GB.
    host_titan = titan and not dp and bool(getattr(config, "TITAN_HOST_GRADIENTS", False))
    if not host_titan:
        # The flat fused optimizer (one rank: no collectives): m / v resident in HBM (or, RAVEN_STATE_ON_HOST, prefetched under the
        # window's last micro-step and written back under the next window), update of the whole flat range in one launch per
        # contiguous trainable range -- the same arithmetic as optimizers.RavenAdamW (which remains the drop-in class for foreign
        # loops and keeps the reference's host residency), without its 0.2 s of exposed host-link time per optimizer step.
        hp = {**_RAVEN_DEFAULTS, **dict(getattr(config, "TITAN_PARAMS" if titan else "RAVEN_PARAMS", {}) or {})}
        curve0 = getattr(config, "LR_CUSTOM_CURVE", [])
        optimizer = (ShardedTitan if titan else ShardedRaven)(
            unet, lr=max(p_[1] for p_ in curve0) if curve0 else config.LEARNING_RATE, betas=tuple(hp["betas"]), eps=hp["eps"],
            weight_decay=hp["weight_decay"], debias_strength=hp["debias_strength"],
            momentum_dtype=_momentum_dtype(hp.get("momentum_dtype", "bfloat16")), clip_grad_norm=float(config.CLIP_GRAD_NORM),
            force_local=not dp,
            # m / v stay resident in HBM (10.3 GB of 288) unless the preset asks for the reference's residency -- pinned host memory,
            # streamed over the host link every optimizer step (raven.py:83-84, 114-117) -- with RAVEN_STATE_ON_HOST = true
            state_on_host=bool(getattr(config, "RAVEN_STATE_ON_HOST", False)))
    else:
        optimizer = _optimizer(config, params)
    flat_opt = isinstance(optimizer, ShardedRaven)
    lr_scheduler = CustomCurveLRScheduler(optimizer, config.LR_CUSTOM_CURVE, config.MAX_TRAIN_STEPS)
    if getattr(config, "RESUME_TRAINING", False):
        if dp:       # every rank resumes its own shard (written next to rank 0's training-state file)
            shard = torch.load(str(config.RESUME_STATE_PATH) + f".rank{rank}", map_location="cpu", weights_only=False)
            ckpt.resume_optimizer(optimizer, shard, lr_scheduler, micro_step)
        else:
            ckpt.resume_optimizer(optimizer, optimizer_state, lr_scheduler, micro_step)

    dataset = feed.CachedLatentDataset(config)
    timestep_sampler = TimestepSampler(config)
    if ts_state is not None:
        timestep_sampler.load_state_dict(ts_state)
    elif getattr(config, "RESUME_TRAINING", False) and micro_step > 0:
        timestep_sampler.set_current_step(micro_step)
    schedule = feed.pack_schedule(feed.batch_schedule(dataset, config.MAX_TRAIN_STEPS, config.BATCH_SIZE, sampler_seed, timestep_sampler.ticket_pool,
                                                     timestep_sampler.bin_ranges, bool(getattr(config, "TIMESTEP_FORCE_IMAGE_BIN_SPREAD", False))),
                                  config.BATCH_SIZE)
    sampler = feed.PrecomputedBatchSampler(schedule, sampler_seed, micro_step if getattr(config, "RESUME_TRAINING", False) else 0)
    loader = torch.utils.data.DataLoader(dataset, batch_sampler=sampler, collate_fn=feed.collate,
                                         num_workers=int(getattr(config, "NUM_WORKERS", 0) if num_workers is None else num_workers))
    sigma_table = None if config.is_rectified_flow else (1.0 - ddpm_alphas_cumprod().float()).clamp_min(0.0).sqrt()
    own_reporter = reporter is None
    if reporter is None:
        reporter = Reporter(config.MAX_TRAIN_STEPS, "conv_in") if rank == 0 else _Silent()
    window = deque(maxlen=GA)
    step_times, optim_times = deque(maxlen=50), deque(maxlen=20)
    t_start = t_last = t_last_opt = time.time()
    noise_gen = torch.Generator()
    clip = float(config.CLIP_GRAD_NORM)
    hist = dict(losses=[], grad_norms=[], lrs=[], saved=[])
    # emergency-save flag: the GUI writes PROJECT_ROOT/force_save.flag and runs the trainer with cwd = PROJECT_ROOT
    # (gui.py:5947, 5983; train.py:2550 looks next to itself) -> default = the process' working directory
    flag = Path(getattr(config, "FORCE_SAVE_FLAG", None) or Path.cwd() / "force_save.flag")
    stem = ckpt.output_model_stem(config, config.SINGLE_FILE_CHECKPOINT_PATH)                # train.py:2334-2349, 2517
    if dp:      # the {uuid} part is random: every rank must write under rank 0's stem
        box = [stem]
        tdist.broadcast_object_list(box, src=0)
        stem = config._RESOLVED_OUTPUT_STEM = box[0]
    unet.zero_grad()
    # Per-micro-step loss read-back WITHOUT draining the queue (the reference blocks on loss.item() every micro-step,
    # train.py:2767): the device scalar is copied (after a scalar all-reduce under data parallel) into a pinned slot behind an
    # event and read LAG = 2 micro-steps LATER, when the next two are already queued (the pinned input staging of TrainStep is double
    # buffered to the same depth): a host hiccup of up to two micro-steps -- a slow DataLoader batch, a busy box -- then costs the GPU
    # nothing (with a lag of one, bench.py's trainer leg showed iterations of 955 / 1040 / 988 / 1264 / 1223 ms on a noisy box where
    # the bare loop, which runs two ahead, held 967).  Only the micro-step that closes an accumulation window is read at once (the
    # optimizer step needs the window mean and the gradient norm anyway).  The reported values and their order are unchanged; a
    # progress line appears up to two micro-steps later than in the reference.
    RING, LAG = 4, 2
    loss_dev = torch.zeros(RING, dtype=torch.float32, device=device)
    loss_host = torch.zeros(RING, dtype=torch.float32).pin_memory()
    loss_ev = [None] * RING
    pending = deque()

    def resolve(rec):
        """Read a queued micro-step's loss, book it, print its progress line."""
        k = rec["slot"]
        loss_ev[k].synchronize()
        v = float(loss_host[k]) / world
        hist["losses"].append(v)
        window.append(v)
        rec["timing"]["loss"] = v
        return v

    norm_host = torch.zeros(RING, dtype=torch.float32).pin_memory()

    def finish_window(rec):
        """The micro-step that closed an accumulation window: its loss, the window mean, the gradient norm of the optimizer step
        (read from the pinned slot its copy landed in) -> the reference's diagnostics record (train.py:2771-2800)."""
        resolve(rec)
        c = rec["closing"]
        raw = c["raw"] if c["raw"] is not None else float(norm_host[rec["slot"]])
        hist["grad_norms"].append(raw)
        hist["lrs"].append(c["lr"])
        diag = dict(optim_step=c["optim_step"], avg_loss=sum(window) / len(window) if window else 0.0, current_lr=c["lr"],
                    raw_grad_norm=raw, clipped_grad_norm=min(raw, clip) if clip > 0 else raw, update_delta=1.0 if raw > 0 else 0.0,
                    optim_step_time=c["optim_step_time"], avg_optim_step_time=c["avg_optim_step_time"])
        window.clear()
        reporter.log_step(rec["micro_step"], timing_data=rec["timing"], diag_data=diag)

    def flush(keep=0):
        while len(pending) > keep:
            rec = pending.popleft()
            if rec.get("closing") is not None:
                finish_window(rec)
                continue
            resolve(rec)
            reporter.log_step(rec["micro_step"], timing_data=rec["timing"], diag_data=None)

    done = False
    gc_frozen = False
    # The loop's CPU tensor work is a handful of tiny ops per micro-step (noise draw, time ids, collate).  With torch's default of one
    # intra-op thread per core every one of them wakes a 256-thread team on the bench boxes, and the op is as slow as the slowest core --
    # on a busy host that was the trainer leg's 80-200 ms hiccups which the bare step (no CPU tensor op in its loop) never saw.
    # Default cap: 8 threads (HOST_THREADS = 0 or >= the current count leaves torch's setting alone).
    host_threads = int(getattr(config, "HOST_THREADS", 8) or 0)
    prev_threads = torch.get_num_threads()
    if 0 < host_threads < prev_threads:
        torch.set_num_threads(host_threads)
    try:
        while not done:
            n_batches = 0
            for batch in loader:
                n_batches += 1
                if micro_step >= config.MAX_TRAIN_STEPS:
                    done = True
                    break
                if not batch:
                    continue
                GB = batch["latents"].shape[0]                   # global micro-batch (after dropped samples)
                micro_step += 1
                diag = None
                timesteps, first_ticket = timestep_sampler.sample(GB)
                noise = generate_noise(batch["latents"], noise_gen, "cpu", step=micro_step, seed=config.SEED)
                jitter = None
                if config.is_rectified_flow:
                    jitter = torch.rand(timesteps.shape, dtype=torch.float32, generator=seeded_torch_generator("cpu", config.SEED, micro_step, 0x5D1))
                    sigma = float(((timesteps[0].float() + jitter[0]) / 1000.0).clamp(0.0, 1.0))
                else:
                    sigma = float(sigma_table[int(timesteps[0])])
                wscale = 1.0
                if dp:                                           # this rank's rows of the global draw (ragged batches: shares differ by <= 1)
                    rows = feed.shard_rows(GB, rank, world)
                    batch = feed.shard_batch(batch, rank, world, even=False)
                    timesteps, noise = timesteps[rows], noise[rows]
                    jitter = jitter[rows] if jitter is not None else None
                    wscale = (rows.stop - rows.start) * world / GB   # every sample weighs 1/GB, as in the single-process run
                latents = batch["latents"]
                B = latents.shape[0]
                tids = make_time_ids(batch.get("scaled_sizes", batch["original_sizes"]), batch.get("crop_coords", [(0, 0)] * B), batch["target_sizes"])
                last = (micro_step % GA == 0)
                # the m / v upload (182 ms of host link at one rank) starts TWO micro-steps (230 ms) before the optimizer step needs it, as in
                # bench.py; started with the last micro-step only, it was hidden or not depending on how far the host happened to run ahead
                # (iterations of 939 and 995-1071 ms in one bench.py trainer leg)
                if flat_opt and (micro_step - 1) % GA == max(0, GA - 2):
                    optimizer.prefetch()
                if B > 0:
                    # HOST tensors go in as they are: TrainStep stages them through its double-buffered pinned area in ONE asynchronous copy
                    # (train_step._host_inputs_in_one_copy).  A `.to(device)` of a pageable tensor here holds the host until the stream has
                    # reached the copy, i.e. until the PREVIOUS micro-step has finished (cProfile: 115 ms per micro-step inside `.to`) -- the
                    # loop then never runs ahead of the GPU and every hiccup of the host idles it
                    loss = step.micro_step(latents, noise, timesteps, batch["embeds"], batch["pooled"], tids, jitter, after_tail=optimizer.reduce_tail if (dp and last and optimizer.overlap) else None,
                                           weight_scale=wscale)
                    slot = micro_step % RING
                    loss_dev[slot:slot + 1].copy_(loss, non_blocking=True)
                else:                                            # fewer samples than ranks: this rank sits the micro-step out
                    slot = micro_step % RING
                    loss_dev[slot:slot + 1].zero_()
                if isinstance(optimizer, TitanAdamW):
                    optimizer.offload_flat(unet)                     # the flat-path form of Titan's post-accumulate hooks
                elif hasattr(optimizer, "accumulate"):
                    optimizer.accumulate()                           # the same under data parallel: fp32 accumulation (dist.ShardedTitan)
                if dp:                                           # reported loss = global mean: sum_r (b_r/GB) * local mean = sum_r loss_r / world
                    tdist.all_reduce(loss_dev[slot:slot + 1])
                loss_host[slot:slot + 1].copy_(loss_dev[slot:slot + 1], non_blocking=True)
                loss_ev[slot] = torch.cuda.Event()
                loss_ev[slot].record()
                now = time.time()
                step_times.append(now - t_last)
                t_last = now
                pending.append(dict(slot=slot, micro_step=micro_step,
                                    timing=dict(raw_step_time=step_times[-1], elapsed_time=now - t_start,
                                                eta=(config.MAX_TRAIN_STEPS - micro_step) * (sum(step_times) / len(step_times)),
                                                loss=0.0, timestep=str(first_ticket), sigma=sigma)))
                flush(keep=LAG)                                  # the loss of the micro-step LAG back (its copy has long landed)
                lr_scheduler.step(micro_step)
                if micro_step % GA == 0:                                                 # train.py:2771-2800
                    cur = pending[-1]
                    raw = None
                    if flat_opt:
                        # [reduce-scatter,] global norm, clip, flat update [, all-gather]: the norm stays on the device and is read with the
                        # closing micro-step's loss, LAG micro-steps later -- the host does not drain the queue at the window's end either
                        # (it did: ~6 ms of idle GPU per iteration while the next batch was fetched and the first micro-step issued)
                        nrm = optimizer.step()
                        norm_host[cur["slot"]:cur["slot"] + 1].copy_(nrm.reshape(1), non_blocking=True)
                        loss_ev[cur["slot"]] = torch.cuda.Event()
                        loss_ev[cur["slot"]].record()             # behind the loss copy AND the norm copy of this slot
                    elif isinstance(optimizer, TitanAdamW):
                        raw = optimizer.clip_grad_norm(clip if clip > 0 else float("inf"))
                        raw = float(raw.item() if isinstance(raw, torch.Tensor) else raw)
                        optimizer.step()
                    else:
                        unet.expose_grads()
                        raw = float(clip_grad_norm_(unet, clip if clip > 0 else float("inf")).item())
                        optimizer.step()
                    optimizer.zero_grad(set_to_none=True)
                    optimizer_step += 1
                    if not gc_frozen:           # the launch tapes / pools built during the first window are permanent: keep the cyclic
                        gc.collect()            # collector from walking them (tens of thousands of objects) in the middle of later windows
                        gc.freeze()
                        gc_frozen = True
                    now = time.time()
                    optim_times.append(now - t_last_opt)
                    t_last_opt = now
                    lr_now = optimizer.param_groups[-1]["lr"]
                    cur["closing"] = dict(raw=raw, lr=lr_now, optim_step=optimizer_step, optim_step_time=optim_times[-1],
                                          avg_optim_step_time=sum(optim_times) / len(optim_times))
                    every = int(getattr(config, "SAVE_EVERY_N_STEPS", 0) or 0)
                    # rank 0 alone consumes the flag file and tells the others: every rank must take the same branch, because
                    # the save path holds collectives (parameter all-gather wait, barrier)
                    forced = ckpt.consume_force_save_flag(flag) if rank == 0 else False
                    if dp:
                        ft = torch.tensor([1 if forced else 0], dtype=torch.int32, device=device)
                        tdist.broadcast(ft, src=0)
                        forced = bool(int(ft.item()))
                    if (every > 0 and optimizer_step % every == 0) or forced:            # train.py:2805-2815
                        reason = "Emergency checkpoint requested" if forced and not (every > 0 and optimizer_step % every == 0) else "Saving checkpoint"
                        flush()                                  # the progress lines of this window come before the checkpoint's (log order of the reference)
                        reporter.log_message(f"\n--- {reason} at optimizer step {optimizer_step} ---")
                        mname, sname = ckpt.checkpoint_names(stem, optimizer_step)
                        if hasattr(optimizer, "synchronize_params"):
                            optimizer.synchronize_params()      # updates / all-gathers still running under the next forward's slots must have landed
                        if dp:
                            Path(config.OUTPUT_DIR).mkdir(parents=True, exist_ok=True)
                            torch.save(optimizer.save_cpu_state(), str(Path(config.OUTPUT_DIR) / sname) + f".rank{rank}")
                        if rank == 0:
                            ckpt.save_model(Path(config.OUTPUT_DIR) / mname, unet, model_to_load, torch.bfloat16)
                            ckpt.save_training_state(Path(config.OUTPUT_DIR) / sname, optimizer_step, micro_step,
                                                     _NoState() if dp else optimizer, sampler.seed, sampler.epoch, timestep_sampler)
                        if dp:
                            tdist.barrier()
                        hist["saved"].append((mname, sname))
                    if not flat_opt:
                        flush()                                  # the module-optimizer paths read their norm on the host: nothing left to wait for
            if n_batches == 0:
                break
        flush()
    finally:
        # an exception or KeyboardInterrupt in the loop must not leave the embedding process (bench legs, tests, foreign callers)
        # with the thread cap or a frozen collector generation
        if torch.get_num_threads() != prev_threads:
            torch.set_num_threads(prev_threads)
        if gc_frozen:
            gc.unfreeze()
    reporter.log_message("\nTraining complete.")
    if own_reporter:
        reporter.shutdown()
    # train.py:2832-2836: the final model, whatever SAVE_EVERY_N_STEPS says
    final = Path(config.OUTPUT_DIR) / f"{stem}.safetensors"
    if hasattr(optimizer, "synchronize_params"):
        optimizer.synchronize_params()           # the last step's overlapped update / all-gather must have landed
    if rank == 0:
        ckpt.save_model(final, unet, model_to_load, torch.bfloat16)
        print("All tasks complete. Final model saved.")
    if dp:
        tdist.barrier()
    hist.update(micro_step=micro_step, optimizer_step=optimizer_step, final_model=str(final))
    return hist


def set_seed(seed):
    """train.py:231-238."""
    import random
    import numpy as np
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    print(f"INFO: Set random seed to {seed}")


def main(argv=None) -> int:
    """Process entry: `python -m aozora_sdxl_training_amd.trainer --config X.json` -- what the GUI spawns for train.py
    (gui.py:5930-5975), here for the native step.  One process per GPU: started by torch.distributed.run (RANK / WORLD_SIZE /
    LOCAL_RANK in the environment) the ranks form an RCCL group and config.BATCH_SIZE is the GLOBAL micro-batch.
    Out of scope, refused with a message instead of being attempted: the Anima DiT mode, offline VAE / text-encoder caching
    (the cache must exist), fp16 mixed precision and paged_adamw_8bit."""
    import os
    import sys
    from .config import TrainingConfig
    config = TrainingConfig(argv)
    if str(config.TRAINING_MODE).lower().startswith("anima"):
        print("ERROR: this build trains the SDXL UNet only; the preset's active mode is Anima DiT.")
        return 2
    if config.MIXED_PRECISION != "bfloat16":
        print(f"ERROR: MIXED_PRECISION={config.MIXED_PRECISION!r}: the HIP step computes in bf16 only.")
        return 2
    if str(config.OPTIMIZER_TYPE).lower() not in ("raven", "titan"):
        raise ValueError(f"Unsupported optimizer type: '{config.OPTIMIZER_TYPE}'")          # train.py:2290
    if config.SEED:
        set_seed(config.SEED)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # this rank's CPUs = the NUMA node of its GPU, intra-op threads capped: before the pinned optimizer state is allocated and
    # before torch starts its intra-op pool (affinity.py; one rank: the mask is left alone)
    from .affinity import bind_rank
    placement = bind_rank(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)), max_threads=int(getattr(config, "HOST_THREADS", 8) or 8))
    if world > 1 and int(os.environ.get("RANK", "0")) == 0:
        print(f"INFO: host placement of rank 0: {placement['n_cpus']} CPUs ({placement['cpus']}), {placement['threads']} threads, {placement['how']}")
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    if world > 1:
        import torch.distributed as tdist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from .streams import host_link_streams
        host_link_streams(device)        # the m / v copy streams take their SDMA engines BEFORE the RCCL communicator exists (streams.py)
        tdist.init_process_group(backend=os.environ.get("AOZORA_DIST_BACKEND", "nccl"), device_id=torch.device(device)
                                 if os.environ.get("AOZORA_DIST_BACKEND", "nccl") == "nccl" else None)
    rank0 = int(os.environ.get("RANK", "0")) == 0
    if rank0:
        if config.RESUME_TRAINING:
            print("\n" + "=" * 50 + "\n--- RESUMING TRAINING SESSION ---\n")
        else:
            print("\n" + "=" * 50 + f"\n--- STARTING {'RECTIFIED FLOW' if config.is_rectified_flow else 'STANDARD SDXL'} TRAINING ---\n" + "=" * 50 + "\n")
        print(f"INFO: Noise type: {config.NOISE_MODE}")
    Path(config.OUTPUT_DIR).mkdir(parents=True, exist_ok=True)
    # --unet-config FILE (JSON of unet_spec.UNetConfig fields): a UNet geometry other than SDXL-base (the test suite's
    # reduced-width model); the GUI never passes it
    import argparse
    import json
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--unet-config", default=None)
    spec = ap.parse_known_args(None if argv is None else list(argv))[0].unet_config
    unet = None
    if spec:
        from .unet_spec import UNetConfig
        with open(spec) as f:
            fields = json.load(f)
        model_cfg = UNetConfig(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in fields.items()})
        src = config.RESUME_MODEL_PATH if config.RESUME_TRAINING else config.SINGLE_FILE_CHECKPOINT_PATH
        unet = ckpt.load_unet(src, device, model_cfg)
    try:
        train(config, unet=unet, device=device)
    finally:
        if world > 1:
            import torch.distributed as tdist
            if tdist.is_initialized():
                tdist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
