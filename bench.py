#!/usr/bin/env python3
"""bench.py -- SDXL UNet train iterations/sec at 1024x1024, global batch 32, on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = ONE training iteration at global batch 32 (BASELINE.json metric): 32/(4*N) micro-steps of
local batch 4 (forward + loss + backward, SDXL-base UNet, 1024^2 => latents 4x128x128, ctx 77x2048),
then gradient reduce-scatter, global-norm clip, Raven AdamW (m/v resident in HBM) and parameter
all-gather.  Global batch is fixed => "scaling": "strong".  Synthetic cached latents / embeddings and
seed-generated weights (no network: no dataset, no SDXL checkpoint).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# One hardware queue per stream of the step (data-gradient, weight-gradient, exchange, m/v H2D, m/v D2H, RCCL's own): with
# the ROCclr default of 4 per priority the 5th stream shares a queue and its kernels serialise behind a 23-190 ms copy
# (aozora_sdxl_training_amd/streams.py).  Read by the HIP runtime at initialisation, i.e. before the first device call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (guides/MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
GLOBAL_BATCH = 32
LOCAL_BATCH = 4
LATENT = 128
PREFETCH_LEAD = int(os.environ.get("AZ_PREFETCH_LEAD", "2"))   # micro-steps before the optimizer step at which the m/v H2D starts
TRAIN_TFLOP_PER_SAMPLE = 20.284   # BASELINE.md section 3 (3 x forward, no recompute)


def init_weights_on_device(unet, seed=1234):
    """Seed-generated weights of the SDXL-base architecture drawn on the device (PyTorch default layer
    init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)); norm weight 1 / bias 0)."""
    g = torch.Generator(device=unet.device).manual_seed(seed)
    shapes = dict(unet._table)
    with torch.no_grad():
        for name, p in unet.named_parameters():
            if ".norm" in name or name.startswith("conv_norm_out"):
                p.fill_(1.0 if name.endswith(".weight") else 0.0)
                continue
            wshape = shapes[name[:-5] + ".weight"] if name.endswith(".bias") else shapes[name]
            bound = 1.0 / math.sqrt(math.prod(wshape[1:]))
            p.copy_(((torch.rand(p.shape, generator=g, device=unet.device) * 2 - 1) * bound).to(torch.bfloat16))


def synthetic_batch(step, micro, rank, B, dev, latent=None, ctx_dim=2048, pooled_dim=1280):
    latent = LATENT if latent is None else latent
    g = torch.Generator().manual_seed(10_000 * step + 100 * micro + rank)
    lat = torch.randn(B, 4, latent, latent, generator=g).bfloat16()
    noise = torch.randn(B, 4, latent, latent, generator=g)
    ctx = torch.randn(B, 77, ctx_dim, generator=g).bfloat16()
    pooled = torch.randn(B, pooled_dim, generator=g).bfloat16()
    tid = torch.tensor([[latent * 8, latent * 8, 0, 0, latent * 8, latent * 8]] * B, dtype=torch.bfloat16)
    ts = torch.randint(0, 1000, (B,), generator=g)
    return [t.to(dev) if t.dtype != torch.int64 else t for t in (lat, noise, ts, ctx, pooled, tid)]


def cpu_baseline(threads, reps=2):
    """the oracle (CPU restatement of train.py:2719-2784 + raven.py:89-149) timed on THIS host on
    BASELINE configs[0]: full-size SDXL-base UNet, epsilon, 512x512 (latent 64x64), batch 1, one whole iteration = forward + loss
    + backward + global-norm clip + Raven AdamW step; 1 warm-up + `reps` timed iterations for the bf16-autocast variant (the
    reference's own dataflow, train.py:273) and for the fp32 variant (the parity oracle).  `value` converts the bf16-autocast
    time to the metric's unit by the BASELINE.md FLOP ratio (4.766 TFLOP for this iteration vs 649.1 for one 1024x1024 gbs-32
    iteration); the raw seconds per cfg1 iteration are reported beside it."""
    from oracle.unet_ref import SDXL_BASE, init_params, forward_macs
    from oracle.step_ref import RefTrainer
    torch.set_num_threads(threads)
    t0 = time.time()
    params = init_params(SDXL_BASE, seed=1234)
    g = torch.Generator().manual_seed(0)
    hw = 64
    lat = torch.randn(1, 4, hw, hw, generator=g).bfloat16()
    noise = torch.randn(1, 4, hw, hw, generator=g)
    ctx = torch.randn(1, 77, 2048, generator=g).bfloat16()
    pooled = torch.randn(1, 1280, generator=g).bfloat16()
    tid = torch.tensor([[512, 512, 0, 0, 512, 512]], dtype=torch.bfloat16)
    ts = torch.tensor([500])
    setup = time.time() - t0
    secs = {}
    for variant, bf16 in (("bf16_autocast", True), ("fp32", False)):
        tr = RefTrainer(SDXL_BASE, params, mode="epsilon", bf16=bf16, ga=1, clip=1.0)      # Raven defaults of config.py:122-129

        def iteration():
            tr.micro_step(lat, noise, ts, ctx, pooled, tid)
            tr.optimizer_step()
        iteration()                                            # warm-up
        t1 = time.time()
        for _ in range(reps):
            iteration()
        secs[variant] = (time.time() - t1) / reps
        del tr
    sample_tflop = 3 * 2 * forward_macs(SDXL_BASE, hw, hw) / 1e12
    iter_tflop = TRAIN_TFLOP_PER_SAMPLE * GLOBAL_BATCH
    dt = secs["bf16_autocast"]
    return dict(value=(sample_tflop / dt) / iter_tflop, unit="iters/sec", cores=threads, kind="port",
                cfg1_seconds_per_iteration=secs, host_cpus=os.cpu_count(), torch_threads=torch.get_num_threads(),
                sample=f"oracle (CPU restatement), BASELINE configs[0]: full SDXL-base UNet, eps, 512x512 (latent 64x64), B=1, one whole "
                       f"iteration = fwd+loss+bwd+clip+Raven step; mean of {reps} after 1 warm-up: bf16-autocast (reference dataflow) "
                       f"{secs['bf16_autocast']:.2f} s/iter = {sample_tflop / secs['bf16_autocast']:.3f} TFLOP/s, fp32 {secs['fp32']:.2f} s/iter = "
                       f"{sample_tflop / secs['fp32']:.3f} TFLOP/s; value = the bf16-autocast rate scaled by FLOPs to one 1024x1024 gbs-32 "
                       f"iteration ({sample_tflop:.3f} -> {iter_tflop:.1f} TFLOP); {threads} threads of {os.cpu_count()} host CPUs; setup {setup:.0f}s")


def live_pmc(cls, timeout_s=150):
    """HBM-side traffic and MFMA-busy share of the dominant class, collected IN THIS RUN: rocprofv3 --pmc passes over
    tools/pmc_target.py (the class' own shapes, 3 launches each) as CHILD processes -- one pass per counter set, as
    MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass).  gfx950 correction: FETCH_SIZE tallies 128-B
    requests at 64 B -> doubled; Infinity-Cache hits are included (fabric-side bytes, an upper bound on HBM bytes).
    Returns None when the profiler is unavailable or a pass fails (the bench never fails on it)."""
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    short = {"gemm_nt": "nt", "gemm_tn": "tn", "conv_fwd": "conv", "conv_dgrad": "conv", "conv_wgrad": "conv", "attn_fwd": "attn", "attn_bwd": "attn"}.get(cls)
    if short is None:
        return None
    work = tempfile.mkdtemp(prefix="az_pmc_")
    env = dict(os.environ, TMPDIR="/tmp", PMC_MANIFEST=os.path.join(work, "manifest.json"))
    merged = {}
    try:
        for tag, counters in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"),
                              ("sq", "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE")):
            d = os.path.join(work, tag)
            cmd = ["timeout", "-k", "10", str(timeout_s), "rocprofv3", "--pmc"] + counters.split() + ["--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.join(ROOT, "tools", "pmc_target.py"), short]
            r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True)
            if r.returncode != 0:
                return dict(error=f"rocprofv3 pass {tag} failed rc={r.returncode}: {(r.stderr or r.stdout)[-300:]}")
            out = os.path.join(work, tag + ".json")
            r2 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_collect.py"), d, env["PMC_MANIFEST"], out], capture_output=True, text=True)
            if r2.returncode != 0:
                return dict(error=f"pmc_collect {tag}: {(r2.stderr or r2.stdout)[-300:]}")
            for sec in json.load(open(out)):
                if sec["cls"] != cls:
                    continue
                m = merged.setdefault(sec["label"], dict(label=sec["label"], calls_per_microstep=sec["calls_per_microstep"],
                                                         algorithmic_bytes_per_launch=sec["algorithmic_bytes_per_launch"], counters={}))
                m["counters"].update(sec["counters"])
    finally:
        shutil.rmtree(work, ignore_errors=True)
    if not merged:
        return None
    shapes = list(merged.values())
    n = sum(x["calls_per_microstep"] for x in shapes)
    for x in shapes:
        c = x["counters"]
        x["hbm_side_bytes_per_launch"] = 2 * c.get("FETCH_SIZE", 0.0) * 1024 + c.get("WRITE_SIZE", 0.0) * 1024
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0             # rocprofv3 sums the 8 XCDs
        x["mfma_busy_frac"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 256 * 4) if gui > 0 else None
    wavg = lambda key: sum(x[key] * x["calls_per_microstep"] for x in shapes if x[key] is not None) / max(n, 1)
    return dict(traffic=wavg("hbm_side_bytes_per_launch"), algorithmic_bytes_per_launch=wavg("algorithmic_bytes_per_launch"),
                mfma_busy_frac=wavg("mfma_busy_frac"), launches_covered=n, shapes=shapes,
                note="rocprofv3 --pmc child passes of this run over tools/pmc_target.py; call-weighted over the class' shapes; "
                     "traffic = 2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes), fabric-side; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 CUs * 4 SIMDs)")


def write_synthetic_cache(root, n_items, latent, ctx_dim, pooled_dim, seed=0):
    """A cache directory in the reference's on-disk format (schema v13: *_lat.pt, *_te.pt, dataset_index.pt, null_embeds.pt --
    train.py:1803-1830, 1966-1986) holding `n_items` random samples of ONE square bucket, for --through-trainer."""
    cache = os.path.join(root, ".precomputed_embeddings_cache_standard_sdxl")
    os.makedirs(cache, exist_ok=True)
    g = torch.Generator().manual_seed(1000 + seed)
    px = latent * 8
    files = []
    for k in range(n_items):
        stem = f"synthetic_{k:04d}"
        meta = dict(relative_path=stem + ".png", original_size=(px, px), scaled_size=(px, px), target_size=(px, px), crop_coords=(0, 0),
                    bucket_variant_index=0)
        lat, te = os.path.join(cache, stem + "_lat.pt"), os.path.join(cache, stem + "_te.pt")
        torch.save({"latents": torch.randn(4, latent, latent, generator=g).bfloat16(), "cache_options": {"cache_schema_version": 13}}, lat)
        torch.save(dict(meta, original_stem=stem, caption_type="txt", caption=f"synthetic {k}",
                        embeds=torch.randn(77, ctx_dim, generator=g).bfloat16(), pooled=torch.randn(pooled_dim, generator=g).bfloat16(),
                        cache_options={"cache_schema_version": 13}), te)
        files.append(dict(meta, te_path=te, lat_path=lat, image_file_signature=None, caption_file_signature=None, caption_signature=None))
    torch.save({"version": 13, "cache_options": {"cache_schema_version": 13}, "files": files}, os.path.join(cache, "dataset_index.pt"))
    torch.save({"embeds": torch.randn(1, 77, ctx_dim, generator=g).bfloat16(), "pooled": torch.randn(1, pooled_dim, generator=g).bfloat16()},
               os.path.join(cache, "null_embeds.pt"))


def through_trainer(unet, dev, world, rank, lb, ga, iters, lat_hw, model_cfg):
    """The same workload through trainer.train -- the loop a user runs: on-disk cache -> DataLoader -> micro-steps -> clip ->
    Raven / sharded Raven -> LR curve -> reporter (loss read back per micro-step).  Returns iterations/sec over the last
    `iters` optimizer steps (the first one is discarded: pools, launch tape)."""
    import shutil
    import tempfile
    import types
    from safetensors.torch import save_file
    from aozora_sdxl_training_amd.trainer import train
    import torch.distributed as dist
    tmp = tempfile.mkdtemp(prefix="az_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None) if rank == 0 else None
    if world > 1:
        box = [tmp]
        dist.broadcast_object_list(box, src=0)
        tmp = box[0]
    try:
        if rank == 0:
            write_synthetic_cache(os.path.join(tmp, "set0"), max(lb * world * 2, 8), lat_hw, model_cfg.cross_attention_dim, model_cfg.pooled_dim)
            save_file({"placeholder.weight": torch.zeros(1)}, os.path.join(tmp, "base.safetensors"))
        if world > 1:
            dist.barrier()
        steps = ga * (iters + 3)        # the first TWO optimizer steps are discarded (pools; the launch tape is recorded on a bucket's second run,
        # which at grad-accum 1 -- eight ranks -- lies inside the second optimizer step) and so is the last (see below)
        cfg = types.SimpleNamespace(
            INSTANCE_DATASETS=[{"path": os.path.join(tmp, "set0"), "repeats": 1}], CAPTION_SOURCE_TYPE="txt", SEED=42, MAX_TRAIN_STEPS=steps,
            BATCH_SIZE=lb * world, GRADIENT_ACCUMULATION_STEPS=ga, PREDICTION_TYPE="epsilon", CLIP_GRAD_NORM=1.0,
            LR_CUSTOM_CURVE=[[0.0, 0.0], [0.05, 8.0e-7], [0.85, 8.0e-7], [1.0, 1.0e-7]], LEARNING_RATE=8e-7, OPTIMIZER_TYPE="raven",
            RAVEN_PARAMS=dict(betas=[0.9, 0.999], eps=1e-8, weight_decay=0.01, debias_strength=0.3, momentum_dtype="bfloat16"),
            UNET_EXCLUDE_TARGETS=[], SAVE_EVERY_N_STEPS=0, OUTPUT_DIR=os.path.join(tmp, "out"), OUTPUT_NAME="bench",
            SINGLE_FILE_CHECKPOINT_PATH=os.path.join(tmp, "base.safetensors"), RESUME_TRAINING=False, TIMESTEP_ALLOCATION=None,
            TIMESTEP_LOSS_WEIGHT_CURVE=None, TIMESTEP_FORCE_IMAGE_BIN_SPREAD=False, NUM_WORKERS=0)

        class Collect:
            def __init__(self): self.t, self.all = [], []
            def log_step(self, micro_step, timing_data=None, diag_data=None):
                self.all.append((micro_step, time.perf_counter(), diag_data is not None))
                if diag_data is not None:       # closing record of an optimizer step, flushed two micro-steps after it was issued
                    self.t.append(time.perf_counter())
            def log_message(self, *a, **k): pass
            def shutdown(self): pass
        col = Collect()
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):          # stdout carries the ONE JSON line only
            train(cfg, unet=unet, device=str(dev), reporter=col)
        if os.environ.get("AZ_BENCH_TRACE"):
            prev = None
            for ms_, t_, opt_ in col.all:
                print(f"[trainer trace] micro-step {ms_}{' (optimizer step)' if opt_ else ''}: +{(t_ - prev) * 1e3 if prev else 0:.1f} ms", file=sys.stderr)
                prev = t_
        if len(col.t) < 2:
            return None
        # The trainer reports a micro-step two micro-steps after it was issued (its loss / gradient norm are read with that lag), so the
        # stamps of the closing records are all late by the same amount -- except the very last one, which the final flush reads as soon
        # as the GPU is done: the last interval is short by that lag and is dropped.
        # The first kept interval would still cover the optimizer step that records the launch tape when grad-accum < 3: dropped as well.
        iters_s = [b - a for a, b in zip(col.t[:-1], col.t[1:])][1:-1]
        if not iters_s:
            return None
        per_iter = sorted(iters_s)
        from aozora_sdxl_training_amd import streams as _streams
        # per-micro-step host timestamps of the measured iterations (what the reporter saw): a host that falls behind shows as
        # uneven micro-steps, a badly placed stream pair as uniformly slow ones
        micro_ms, prev = [], None
        for ms_, t_, opt_ in col.all:
            if prev is not None:
                micro_ms.append(round((t_ - prev) * 1e3, 1))
            prev = t_
        return dict(iters_per_sec=1.0 / per_iter[len(per_iter) // 2],          # median iteration (one hiccup of the host does not decide a 3-5 iteration leg)
                    iteration_ms=[round(x * 1e3, 1) for x in iters_s], micro_step_ms=micro_ms[-ga * min(len(iters_s), 2):],
                    stream_probe=list(_streams.log))
    finally:
        if world > 1:
            dist.barrier()
        if rank == 0:
            shutil.rmtree(tmp, ignore_errors=True)


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` started bare (no torch.distributed.run around it): this process has not touched the GPU (importing
    torch does not) and becomes the launcher -- N fresh child processes, one rank per GPU, rendezvous on 127.0.0.1; rank 0's
    stdout (the ONE JSON line) is relayed, everything else goes to stderr; exit code = the worst child's.  Nothing is exec-replaced."""
    import socket
    import subprocess
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    # rank 0's stdout is drained by a reader thread while ALL children are polled: the first non-zero exit (a rank that died before
    # or during rendezvous) tears the others down at once instead of leaving them to sit out the collective / store timeout, and
    # an overall limit (BENCH_SPAWN_TIMEOUT seconds, default 3600) bounds the wait
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + float(os.environ.get("BENCH_SPAWN_TIMEOUT", "3600"))
    failed = None
    while True:
        codes = [p_.poll() for p_ in procs]
        if all(c is not None for c in codes):
            break
        bad = [(i, c) for i, c in enumerate(codes) if c not in (None, 0)]
        if bad or time.monotonic() > deadline:
            failed = f"rank {bad[0][0]} exited with code {bad[0][1]}" if bad else "overall timeout"
            print(f"[bench] {failed}: stopping the other ranks", file=sys.stderr)
            for p_ in procs:
                if p_.poll() is None:
                    p_.terminate()
            t_end = time.monotonic() + 10
            for p_ in procs:
                try:
                    p_.wait(timeout=max(0.1, t_end - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p_.kill(); p_.wait()
            break
        time.sleep(0.2)
    reader.join(timeout=5)
    out0 = buf[0] if buf else ""
    rcs = [p_.returncode for p_ in procs]
    if failed and all(c == 0 for c in rcs):
        rcs[0] = 124                                 # timeout with every child reaped cleanly: still a failure
    for line in (out0 or "").splitlines():          # only the JSON line belongs on stdout (gloo's C++ side prints its own chatter there)
        print(line, file=sys.stdout if line.lstrip().startswith("{") else sys.stderr)
    sys.stdout.flush()
    worst = max(rcs, key=lambda c: abs(c))
    if worst != 0:
        print(f"[bench] rank exit codes: {rcs}", file=sys.stderr)
    return worst


TFLOP_PER_SAMPLE_BY_LATENT = {64: 4.766, 96: 10.924, 112: 15.162, 128: 20.284}     # BASELINE.md section 3 / SURVEY 8d
FROZEN_WGRAD_TFLOP_PER_SAMPLE_CFG5 = 0.797        # mid_block's weight gradients at 1024^2 (0.399 TMAC forward)
LOGIT_NORMAL = {"bin_size": 100, "counts": [45, 143, 176, 173, 154, 126, 94, 59, 26, 4]}      # SURVEY 8c F5 (mu -0.5, sigma 1)
LEGS = {
    "cfg2": "BASELINE configs[1]: epsilon, 1024^2, B=4 x GA 8, Raven (the headline workload)",
    "cfg3": "BASELINE configs[2], one rank's kernels at global batch 32 on ONE GPU: v_prediction, logit-normal timestep tickets, 1024^2, B=4 x GA 8, Raven",
    "cfg4": "BASELINE configs[3], one rank's workload: rectified_flow (ticket + seeded jitter), buckets 768^2 / 896^2 / 1024^2 cycling per "
            "micro-step, B=4 x GA 4, Raven (three activation pools / launch tapes)",
    "cfg5_titan_host": "BASELINE configs[4] on ONE GPU: freeze keywords 'mid_block, up_blocks.3' (413 M parameters, their weight gradients "
                       "elided), v_prediction + tickets, 1024^2, B=4 x GA 8, optimizers.TitanAdamW -- the reference's residency: fp32 "
                       "gradients in pinned HOST memory, written over the host link after every micro-step (titan.py:119-131)",
    "cfg5_titan_device": "the same with dist.ShardedTitan(force_local): fp32 gradient accumulator in HBM, same arithmetic (titan.py:162-184, 230-296)",
    "cfg2_state_on_host": "the headline workload with the REFERENCE's residency of the Raven moments (raven.py:83-84, 114-117): m / v in pinned host "
                          "memory, 20.5 GB streamed over the host link per optimizer step (dist.ShardedRaven(state_on_host=True)); the headline "
                          "keeps the 10.3 GB resident in HBM -- same kernels on the same values",
    "lb8": "cfg2 with local batch 8 x GA 4 (same global batch 32; SURVEY 8d allows a larger local batch if memory allows -- never the headline)",
    "lb16": "cfg2 with local batch 16 x GA 2",
}


def run_leg(name, iters, dev):
    """One secondary workload on cuda:0, in its own process (`bench.py --leg NAME`): 2 warm-up iterations, `iters` timed ones
    between two drains of the compute streams; inputs resident in HBM.  Returns ms per iteration and model TFLOP/s (weight gradients of frozen
    layers subtracted)."""
    from aozora_sdxl_training_amd.unet import AozoraUNet
    from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
    from aozora_sdxl_training_amd.train_step import TrainStep
    from aozora_sdxl_training_amd.dist import ShardedRaven, ShardedTitan
    from aozora_sdxl_training_amd.schedule import build_timestep_ticket_pool, trainable_mask, seeded_torch_generator
    lb = {"lb8": 8, "lb16": 16}.get(name, LOCAL_BATCH)
    ga = 4 if name == "cfg4" else GLOBAL_BATCH // lb
    mode = {"cfg3": "v_prediction", "cfg4": "rectified_flow", "cfg5_titan_host": "v_prediction", "cfg5_titan_device": "v_prediction"}.get(name, "epsilon")
    latents = [96, 112, 128] if name == "cfg4" else [LATENT]
    unet = AozoraUNet(SDXL_BASE, dev)
    init_weights_on_device(unet)
    frozen = 0
    if name.startswith("cfg5"):
        names = [n for n, _ in unet.named_parameters()]
        for (n, p), m in zip(unet.named_parameters(), trainable_mask(names, ["mid_block", "up_blocks.3"])):
            p.requires_grad = m
            frozen += 0 if m else p.numel()
    step = TrainStep(unet, mode=mode, grad_accum=ga, world_size=1, use_graph=False)
    hp = dict(lr=8e-7, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, debias_strength=0.3, momentum_dtype=torch.bfloat16)
    host_titan = name == "cfg5_titan_host"
    if host_titan:
        from aozora_sdxl_training_amd.optimizers import TitanAdamW
        opt = TitanAdamW([{"params": [p for p in unet.parameters() if p.requires_grad], "lr_scale": 1.0}], **hp)
    elif name == "cfg5_titan_device":
        opt = ShardedTitan(unet, clip_grad_norm=1.0, force_local=True, **hp)
    else:
        opt = ShardedRaven(unet, clip_grad_norm=1.0, state_on_host=(name == "cfg2_state_on_host"), **hp)
    total_micro = ga * (iters + 2)
    tickets = None
    if name in ("cfg3", "cfg5_titan_host", "cfg5_titan_device"):
        tickets, _ = build_timestep_ticket_pool(LOGIT_NORMAL, total_micro * lb, 1000, 42, False)
    # one resident batch per bucket (inputs are not part of the timed path); timesteps / jitter change per micro-step
    batches = {hw: synthetic_batch(0, i, 0, lb, dev, hw) for i, hw in enumerate(latents)}
    jit0 = torch.full((lb,), 0.5) if mode == "rectified_flow" else None
    for hw in latents:                               # pool / launch-tape phase per bucket, gradients discarded
        for _ in range(3):
            step.micro_step(*batches[hw], jit0)
    step.synchronize()
    unet.zero_grad()
    seq = []
    ms_ = [0]

    def iteration():
        for m in range(ga):
            ms_[0] += 1
            hw = latents[(ms_[0] - 1) % len(latents)]
            seq.append(hw)
            lat, noise, ts, ctx, pooled, tid = batches[hw]
            jit = None
            if tickets is not None:
                ts = torch.tensor(tickets[(ms_[0] - 1) * lb: ms_[0] * lb])
            if mode == "rectified_flow":
                jit = torch.rand(ts.shape, dtype=torch.float32, generator=seeded_torch_generator("cpu", 42, ms_[0], 0x5D1))
            if not host_titan and m == max(0, ga - PREFETCH_LEAD):
                opt.prefetch()
            step.micro_step(lat, noise, ts, ctx, pooled, tid, jit)
            if host_titan:
                opt.offload_flat(unet)               # trainer.train: the flat-path form of Titan's post-accumulate hooks
            elif hasattr(opt, "accumulate"):
                opt.accumulate()
        if host_titan:
            opt.clip_grad_norm(1.0)
            opt.step()
        else:
            opt.step()
        opt.zero_grad(set_to_none=True)

    def drain_compute():
        """Wait for the COMPUTE streams only (data-gradient stream, parameter-gradient stream(s), the caller's stream).  The closing
        m / v write-back of an optimizer step runs on a copy stream under the NEXT iteration (190 ms of host link at one rank): a
        device-wide synchronize behind K = 2-3 iterations would charge each of them 60-95 ms that the steady state does not pay
        (the headline's K = 20 charges 9.5 ms)."""
        step.synchronize()
        for sd in unet._sides:
            sd.synchronize()
        torch.cuda.current_stream().synchronize()

    WARM = 2
    for _ in range(WARM):
        iteration()
    drain_compute()
    seq.clear()
    t0 = time.perf_counter()
    for _ in range(iters):
        iteration()
    drain_compute()
    dt = (time.perf_counter() - t0) / iters
    torch.cuda.synchronize()
    tflop = sum(TFLOP_PER_SAMPLE_BY_LATENT[hw] - (FROZEN_WGRAD_TFLOP_PER_SAMPLE_CFG5 if frozen else 0.0) for hw in seq) * lb / iters
    return dict(what=LEGS[name], ms_per_iteration=dt * 1e3, micro_steps_per_iteration=ga, local_batch=lb, ms_per_micro_step=dt * 1e3 / ga,
                samples_per_sec=lb * ga / dt, model_tflop_per_iteration=tflop, model_tflops=tflop / dt,
                mfma_roofline_frac=tflop / dt / PEAK_BF16_TFLOPS, iterations_timed=iters, warmup=WARM,
                timing="steady state: K iterations between two drains of the compute streams; the last optimizer step's m / v write-back "
                       "(copy stream, hidden under the next iteration) is outside the bracket",
                frozen_parameters=frozen or None, hbm_reserved_gib=torch.cuda.memory_reserved(dev) / 2 ** 30)


def other_configs(names, iters, budget_s):
    """Each leg in a fresh child process (its own streams, pools and HBM), one after the other, within an overall time budget;
    a leg that fails or is skipped is reported as such -- the headline never depends on it."""
    import subprocess
    out, t0 = {}, time.monotonic()
    for name in names:
        left = budget_s - (time.monotonic() - t0)
        if left < 45:
            out[name] = dict(what=LEGS[name], skipped=f"time budget of {budget_s:.0f} s for the secondary legs spent")
            continue
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg", name, "--leg-iters", str(iters)], capture_output=True,
                               text=True, timeout=min(left, 240))
            out[name] = json.loads(r.stdout.strip().splitlines()[-1])["leg"] if r.returncode == 0 else dict(what=LEGS[name], failed=f"rc={r.returncode}: {r.stderr[-300:]}")
        except Exception as e:
            out[name] = dict(what=LEGS[name], failed=repr(e))
    return out


def hbm_roofline(breakdown):
    """The HBM-bound kernel classes of the profiled micro-step (normalisation, GEGLU, element-wise) against the 8 TB/s peak:
    algorithmic bytes of the class (bytes per element as DESIGN.md section 4 states them) / its serialised HIP-event time."""
    out = []
    for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1]["ms"]):
        if v["flops"] > 0 or v["bytes"] <= 0 or v["ms"] <= 0:
            continue
        gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9
        out.append(dict(kernel=k, calls_per_microstep=v["calls"], ms_per_microstep=v["ms"], achieved=gbs, peak=PEAK_HBM_GBS, unit="GB/s",
                        frac=gbs / PEAK_HBM_GBS, avg_launch_us=v["ms"] / v["calls"] * 1e3))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay a captured hipGraph per micro-step instead of eager multi-stream issue")
    ap.add_argument("--profile-out", default=None, help="write the per-op-class event breakdown here (json)")
    ap.add_argument("--local-batch", type=int, default=LOCAL_BATCH,
                    help="samples per micro-step and GPU (BASELINE configs[1] = 4; larger values are an experiment: same global batch, fewer micro-steps)")
    ap.add_argument("--serial", action="store_true", help="issue everything on one stream (for profiles whose per-kernel durations are uncontended)")
    ap.add_argument("--double-buffer", action="store_true", help="experiment: two activation pools, deferred weight-gradient join")
    ap.add_argument("--through-trainer", type=int, default=5, metavar="ITERS",
                    help="also time ITERS iterations of the same workload through trainer.train (on-disk cache, DataLoader, reporter); 0 = skip")
    ap.add_argument("--no-live-pmc", action="store_true", help="skip the rocprofv3 --pmc child passes (traffic / MFMA-busy of the dominant class)")
    ap.add_argument("--trainer-child", type=int, default=0, help=argparse.SUPPRESS)     # internal: the --through-trainer leg in its own process
    ap.add_argument("--config", default="cfg2", choices=sorted(LEGS),
                    help="cfg2 (default) = the headline line; any other name times THAT workload alone on one GPU and prints {'leg': ...}")
    ap.add_argument("--leg", default=None, choices=sorted(LEGS), help=argparse.SUPPRESS)   # internal: one secondary leg in its own process
    ap.add_argument("--leg-iters", type=int, default=3, help="timed iterations per secondary leg (after 2 warm-ups)")
    ap.add_argument("--other-configs", default="all", choices=["all", "none"],
                    help="after the headline measurement (N = 1): also time cfg3 / cfg4 / cfg5 (host and device Titan) and local batches 8 / 16, "
                         "each in a child process, reported under 'other_configs' / 'local_batch_sweep'")
    ap.add_argument("--state-on-host", action="store_true", help="A/B: the reference's residency of the Raven moments (pinned host memory, streamed "
                    "per optimizer step) for the headline run; the line then says so in config.workload")
    ap.add_argument("--other-budget", type=float, default=330.0, help="seconds the secondary legs may take in total")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="N>1 rehearsal on a box with ONE GPU: all ranks share cuda:0 and exchange through gloo (exercises the "
                         "data-parallel control flow of this script; the production backend is nccl = RCCL)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:      # started bare: be the launcher (no GPU call has been made in this process)
        raise SystemExit(spawn_ranks(a.gpus, sys.argv[1:]))
    leg = a.leg or (a.config if a.config != "cfg2" else None)
    if leg is not None:                                    # one secondary workload, one GPU, one JSON object
        if a.gpus != 1:
            raise SystemExit("--config / --leg other than cfg2 time one GPU's workload: use --gpus 1")
        torch.cuda.set_device(0)
        print(json.dumps({"leg": dict(run_leg(leg, max(1, a.leg_iters), torch.device("cuda", 0)), name=leg)}), flush=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: start it bare (python bench.py --gpus N) or under torch.distributed.run --nproc-per-node N")
    import torch.distributed as dist
    # this rank's CPUs (the NUMA node of its GPU) and its intra-op thread cap: BEFORE the pinned m / v shards are allocated and
    # before torch starts an intra-op pool (8 ranks x 128-256 threads otherwise); no wrapper process (affinity.py)
    from aozora_sdxl_training_amd.affinity import bind_rank
    placement = bind_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), max_threads=int(os.environ.get("AZ_HOST_THREADS", "8")))
    if a.rehearse_gloo:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # the m / v copy streams make their first pinned copies (= take their SDMA engines) BEFORE the RCCL communicator exists: created
        # after it, the H2D of the optimizer state runs as blit kernels and the micro-steps beside it take 137-141 ms instead of 116
        # (aozora_sdxl_training_amd/streams.py host_link_streams; profiles/r04_host_link_and_rccl.txt)
        from aozora_sdxl_training_amd.streams import host_link_streams
        host_link_streams(dev)
        if a.rehearse_gloo:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    from aozora_sdxl_training_amd import ops
    from aozora_sdxl_training_amd.unet import AozoraUNet
    from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
    from aozora_sdxl_training_amd.train_step import TrainStep
    from aozora_sdxl_training_amd.dist import ShardedRaven

    lb = a.local_batch
    assert GLOBAL_BATCH % (lb * world) == 0
    ga = GLOBAL_BATCH // (lb * world)
    if a.rehearse_gloo:        # control-flow rehearsal only: mini SDXL-topology UNet, 128x128 px (the printed value is meaningless)
        from aozora_sdxl_training_amd.unet_spec import mini_config
        model_cfg, lat_hw = mini_config(), 16
    else:
        model_cfg, lat_hw = SDXL_BASE, LATENT
    unet = AozoraUNet(model_cfg, dev)
    unet.concurrent_wgrad = not a.serial      # --serial: one stream from the very first launch (pool phase, warm-up, timed region)
    init_weights_on_device(unet)
    if a.trainer_child > 0:        # a fresh process: the trainer creates its streams in its usual order (streams.py: a stream that lands
        # on a hardware queue shared with the copy streams serialises behind 180 ms host-link copies -- measured 0.45 it/s when the
        # trainer ran inside the bench process beside the bench's own optimizer streams)
        v = through_trainer(unet, dev, world, rank, lb, ga, a.trainer_child, lat_hw, model_cfg)
        if rank == 0:
            print(json.dumps({"trainer": v}), flush=True)
        return
    # experiment (measured neutral: 0.851 vs 0.852 it/s -- the step is throughput-bound, the forward has no idle capacity
    # to absorb deferred weight-gradient work): two activation pools, non-final micro-steps do not join their wgrad branch
    dbuf = (ga > 1) and not a.graph and a.double_buffer
    step = TrainStep(unet, mode="epsilon", grad_accum=ga, world_size=world, use_graph=a.graph, double_buffer=dbuf)
    opt = ShardedRaven(unet, lr=8e-7, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, debias_strength=0.3,
                       momentum_dtype=torch.bfloat16, clip_grad_norm=1.0, state_on_host=a.state_on_host)
    # one fixed set of synthetic micro-batches resident in HBM (inputs are not part of the timed path)
    batches = [synthetic_batch(0, m, rank, lb, dev, lat_hw, model_cfg.cross_attention_dim, model_cfg.pooled_dim)
               for m in range(min(ga, 2))]

    def iteration():
        losses = []
        for m in range(ga):
            if m == max(0, ga - PREFETCH_LEAD):
                opt.prefetch()        # m/v H2D rides under the last micro-steps (they do not depend on the gradients)
            # last micro-step of the window: the tail region's reduce-scatter starts right after the mid block's backward
            hook = opt.reduce_tail if (m == ga - 1 and opt.overlap and not a.graph) else None
            losses.append(step.micro_step(*batches[m % len(batches)], after_tail=hook, defer_join=dbuf and m < ga - 1))
        gn = opt.step()
        opt.zero_grad(set_to_none=True)
        return losses[-1], gn

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # "compile" phase (not a training step, outside --warmup and the timed region): run 0 allocates the activation
    # pool; with --graph run 1 captures the hipGraph and run 2 replays it; gradients are discarded.
    t_c = time.perf_counter()
    for _ in range(3):
        step.micro_step(*batches[0])
    step.synchronize()
    opt.zero_grad(set_to_none=True)
    if rank == 0:
        print(f"[bench] pool allocation / graph capture phase: {time.perf_counter() - t_c:.1f} s", file=sys.stderr, flush=True)
        from aozora_sdxl_training_amd import streams as _streams
        for line in _streams.log:
            print(f"[bench] stream choice: {line}", file=sys.stderr, flush=True)

    for _ in range(a.warmup):
        iteration()
    opt.enable_timing()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss, gn = iteration()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    loss_v, gn_v = float(loss.item()), float(gn.item())
    # exchange anatomy of the timed iterations, per rank (events on the streams the pieces ran on): optimizer boundary as the
    # main stream saw it (what is NOT hidden), per-region collective times / rates, m / v host-link copies
    exch = opt.timing_summary()
    opt.enable_timing(False)
    exch = dict(exch or {}, host_placement=placement)
    exch_all = [exch]
    if world > 1:
        exch_all = [None] * world
        dist.all_gather_object(exch_all, exch)

    # ---- per-kernel roofline: one eager micro-step bracketed launch-by-launch with HIP events ----
    roof, breakdown = None, None
    if rank == 0:
        ops.PROFILER = ops.Profiler()
        unet.concurrent_wgrad = False        # serialise the two backward branches so per-launch durations are uncontended
        prof_step = TrainStep(unet, mode="epsilon", grad_accum=ga, world_size=world, use_graph=False)
        prof_step.micro_step(*batches[0])        # populates this step object's pools (events included, discarded)
        prof_step.synchronize()
        ops.PROFILER.summary()
        prof_step.micro_step(*batches[0])
        prof_step.synchronize()
        breakdown = ops.PROFILER.summary()
        ops.PROFILER = None
        unet.concurrent_wgrad = not a.serial
        unet.zero_grad()
        tot_ms = sum(v["ms"] for v in breakdown.values())
        dom = max(breakdown.items(), key=lambda kv: kv[1]["ms"])
        k, v = dom
        if v["flops"] > 0:
            ach = v["flops"] / v["calls"] / (v["ms"] / v["calls"] * 1e-3) / 1e12
            roof = dict(kernel=k, bound="mfma", achieved=ach, peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_BF16_TFLOPS,
                        traffic=None, calls_per_microstep=v["calls"], avg_launch_ms=v["ms"] / v["calls"],
                        share_of_microstep=v["ms"] / tot_ms)
        else:
            ach = v["bytes"] / v["calls"] / (v["ms"] / v["calls"] * 1e-3) / 1e9
            roof = dict(kernel=k, bound="hbm", achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s", frac=ach / PEAK_HBM_GBS,
                        traffic=None, calls_per_microstep=v["calls"], avg_launch_ms=v["ms"] / v["calls"],
                        share_of_microstep=v["ms"] / tot_ms)
        roof["algorithmic_bytes_per_launch"] = v["bytes"] / v["calls"]
        # HBM-side traffic per launch and MFMA-busy share of the dominant class, collected in THIS run by rocprofv3 --pmc child
        # passes (separate passes per counter set, gfx950 FETCH_SIZE correction: MI355X_MICROARCH.md); null when the profiler cannot run here
        pm = None if a.no_live_pmc else live_pmc(k)
        if pm is not None and "error" not in pm:
            roof["traffic"] = pm["traffic"]
            roof["traffic_unit"] = "bytes/launch, fabric-side (live rocprofv3 --pmc passes of this run, call-weighted over %d of %d launches per micro-step)" % (pm["launches_covered"], v["calls"])
            roof["traffic_algorithmic_bytes_per_launch_same_shapes"] = pm["algorithmic_bytes_per_launch"]
            roof["mfma_busy_frac"] = pm["mfma_busy_frac"]
            roof["pmc_shapes"] = [dict(label=x["label"], calls=x["calls_per_microstep"], hbm_side_bytes=x["hbm_side_bytes_per_launch"],
                                       algorithmic_bytes=x["algorithmic_bytes_per_launch"], mfma_busy_frac=x["mfma_busy_frac"]) for x in pm["shapes"]]
        else:          # no committed stand-in: a traffic figure is either measured in this run or absent
            roof["traffic_unit"] = "unavailable in this run (live rocprofv3 --pmc passes: %s)" % ((pm or {}).get("error", "skipped"),)
        if a.profile_out:
            with open(a.profile_out, "w") as f:
                json.dump(dict(microstep_ms_eager=tot_ms, classes=breakdown), f, indent=1)

    # ---- the same workload through trainer.train (the loop a user runs) -------------------------------------------------
    trainer_its = None
    run_trainer = a.through_trainer > 0 and not a.graph and not a.rehearse_gloo and world == 1
    run_others = world == 1 and a.other_configs == "all" and not a.graph and not a.rehearse_gloo
    if run_trainer or run_others:
        opt.synchronize_state()
        del opt, step, batches, unet          # give the device memory back: every child builds its own model
        import gc
        gc.collect(); torch.cuda.empty_cache()
    if run_trainer:
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", "1", "--trainer-child", str(a.through_trainer),
                                "--local-batch", str(lb)], capture_output=True, text=True, timeout=600)
            trainer_its = json.loads(r.stdout.strip().splitlines()[-1])["trainer"] if r.returncode == 0 else f"failed rc={r.returncode}: {r.stderr[-200:]}"
        except Exception as e:          # reported beside the measurement; never fails the bench
            trainer_its = f"failed: {e!r}"

    others = None
    if run_others:
        others = other_configs(["cfg3", "cfg4", "cfg5_titan_device", "cfg5_titan_host", "cfg2_state_on_host", "lb8", "lb16"], a.leg_iters, a.other_budget)

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            cpu = cpu_baseline(min(len(os.sched_getaffinity(0)), 64))
        except Exception as e:   # the baseline is reported beside the measurement; never fail the bench on it
            cpu = dict(value=None, unit="iters/sec", cores=0, kind="port", sample=f"failed: {e!r}")

    if rank == 0:
        its = a.steps / dt
        out = {
            "metric": "SDXL UNet train iters/sec (1024px, gbs=32)", "value": its, "unit": "iters/sec", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "SDXL-base UNet (2.567B params), epsilon pred, 1024x1024 (latent 4x128x128), ctx 77x2048, "
                                   f"local batch {lb} x grad-accum {ga} x {world} GPU = global batch 32, Raven AdamW (" + ("bf16 m/v in pinned host memory, streamed per step (--state-on-host)" if a.state_on_host else
                                   "bf16 m/v resident in HBM -- 10.3 GB of 288; the reference parks them in pinned host memory to fit 24 GB cards: "
                                   "other_configs.cfg2_state_on_host --") + ", sharded 1/N per rank), clip 1.0, " + ("hipGraph replay" if a.graph else "eager 2-stream issue (dgrad chain || wgrad branch)"),
                       "global_batch": GLOBAL_BATCH, "parallelism": f"dp{world}"},
            "model_tflops_per_gpu": TRAIN_TFLOP_PER_SAMPLE * GLOBAL_BATCH / world * its,
            "mfma_roofline_frac_whole_step": TRAIN_TFLOP_PER_SAMPLE * GLOBAL_BATCH / world * its / PEAK_BF16_TFLOPS,
            "last_loss": loss_v, "last_grad_norm": gn_v,
            "roofline": roof, "hbm_roofline": hbm_roofline(breakdown) if breakdown else None, "cpu_baseline": cpu,
            "through_trainer": None if trainer_its is None else dict(
                value=trainer_its["iters_per_sec"] if isinstance(trainer_its, dict) else trainer_its, unit="iters/sec",
                vs_value=(trainer_its["iters_per_sec"] / its) if isinstance(trainer_its, dict) and its else None,
                iteration_ms=trainer_its.get("iteration_ms") if isinstance(trainer_its, dict) else None,
                stream_probe=trainer_its.get("stream_probe") if isinstance(trainer_its, dict) else None,
                # the per-micro-step trace rides along when the loop is more than 3 % off the bare step
                micro_step_ms=trainer_its.get("micro_step_ms") if isinstance(trainer_its, dict) and its and trainer_its["iters_per_sec"] < 0.97 * its else None,
                what=f"trainer.train on the same workload (synthetic on-disk cache -> DataLoader -> micro-steps -> clip -> Raven -> "
                     f"reporter, loss read back per micro-step with a lag of two), median of {a.through_trainer} optimizer steps (the first two and the last of {a.through_trainer + 3} discarded)"),
            "exchange": dict(per_rank=exch_all, note="mean ms per optimizer step over the timed iterations, HIP events on the stream each "
                             "piece ran on: optimizer_boundary_on_main_stream = what the step adds to the main stream (not hidden); "
                             "reduce_scatter / all_gather per region with their GB/s (payload bytes of the region / time); mv_h2d / mv_d2h (only with "
                             "state_on_host) = the owned shard of the pinned host Raven state over the host link"),
        }
        if others is not None:
            out["other_configs"] = {k: v for k, v in others.items() if k.startswith("cfg")}
            out["local_batch_sweep"] = dict({k: v for k, v in others.items() if k.startswith("lb")},
                                            note="secondary: the same global batch 32 with a larger local batch (fewer, larger micro-steps); "
                                                 "never the headline, which stays BASELINE configs[1] (local batch 4)")
        if a.rehearse_gloo:
            out["rehearsal"] = "mini UNet over gloo on one GPU: control-flow check only, the numbers are meaningless"
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()                    # rank 0 was still profiling: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
