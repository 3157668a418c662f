#!/usr/bin/env python3
"""bench.py -- SDXL UNet train iterations/sec at 1024x1024, global batch 32, on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = ONE training iteration at global batch 32 (BASELINE.json metric): 32/(4*N) micro-steps of
local batch 4 (forward + loss + backward, SDXL-base UNet, 1024^2 => latents 4x128x128, ctx 77x2048),
then gradient reduce-scatter, global-norm clip, Raven AdamW (m/v in pinned host memory) and parameter
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


def cpu_baseline(threads):
    """The oracle (CPU restatement of train.py:2719-2784, fp32 PyTorch ops) timed on this host: full-size
    SDXL-base UNet, one sample at 256x256 px (latent 32x32), fwd+loss+bwd; scaled to the metric's unit by
    the BASELINE.md FLOP ratio.  Bounded sample so the default run stays within minutes."""
    from oracle.unet_ref import SDXL_BASE, init_params, forward_macs
    from oracle.step_ref import RefTrainer
    torch.set_num_threads(threads)
    t0 = time.time()
    params = init_params(SDXL_BASE, seed=1234)
    tr = RefTrainer(SDXL_BASE, params, mode="epsilon", bf16=False, ga=1)
    del params
    g = torch.Generator().manual_seed(0)
    hw = 32
    lat = torch.randn(1, 4, hw, hw, generator=g).bfloat16()
    noise = torch.randn(1, 4, hw, hw, generator=g)
    ctx = torch.randn(1, 77, 2048, generator=g)
    pooled = torch.randn(1, 1280, generator=g)
    tid = torch.tensor([[256, 256, 0, 0, 256, 256]], dtype=torch.bfloat16)
    ts = torch.tensor([500])
    setup = time.time() - t0
    tr.micro_step(lat, noise, ts, ctx, pooled, tid)        # warm-up
    for p in tr.params.values():
        p.grad = None
    t1 = time.time()
    for _ in range(2):                                     # 2 timed repetitions after the warm-up
        tr.micro_step(lat, noise, ts, ctx, pooled, tid)
        for p in tr.params.values():
            p.grad = None
    dt = (time.time() - t1) / 2
    sample_tflop = 3 * 2 * forward_macs(SDXL_BASE, hw, hw) / 1e12
    iter_tflop = TRAIN_TFLOP_PER_SAMPLE * GLOBAL_BATCH
    return dict(value=(sample_tflop / dt) / iter_tflop, unit="iters/sec", cores=threads, kind="port",
                sample=f"oracle fp32, full SDXL-base UNet, 1 sample @256x256px (latent 32x32), fwd+loss+bwd {dt:.2f}s (mean of 2 after 1 warm-up) = "
                       f"{sample_tflop / dt:.3f} TFLOP/s, scaled by FLOPs to one 1024x1024 gbs-32 iteration ({iter_tflop:.1f} TFLOP); setup {setup:.0f}s")


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
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="N>1 rehearsal on a box with ONE GPU: all ranks share cuda:0 and exchange through gloo (exercises the "
                         "data-parallel control flow of this script; the production backend is nccl = RCCL)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    import torch.distributed as dist
    if a.rehearse_gloo:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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
    # experiment (measured neutral: 0.851 vs 0.852 it/s -- the step is throughput-bound, the forward has no idle capacity
    # to absorb deferred weight-gradient work): two activation pools, non-final micro-steps do not join their wgrad branch
    dbuf = (ga > 1) and not a.graph and a.double_buffer
    step = TrainStep(unet, mode="epsilon", grad_accum=ga, world_size=world, use_graph=a.graph, double_buffer=dbuf)
    opt = ShardedRaven(unet, lr=8e-7, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, debias_strength=0.3,
                       momentum_dtype=torch.bfloat16, clip_grad_norm=1.0)
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
        # HBM-side traffic per launch of the dominant class: from the committed rocprofv3 --pmc passes (bench.py cannot
        # run the profiler on itself); call-weighted over the class' own shapes, gfx950-corrected (see the file's note)
        pmc_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"r01_e_pmc_{k}.json")
        if os.path.exists(pmc_file):
            with open(pmc_file) as f:
                pm = json.load(f)
            roof["traffic"] = pm["mean_hbm_bytes_per_launch"]
            roof["traffic_unit"] = "bytes/launch (PMC, call-weighted over %d of %d launches)" % (pm["calls_covered"], pm["calls_in_class"])
            roof["algorithmic_bytes_per_launch"] = v["bytes"] / v["calls"]
        if a.profile_out:
            with open(a.profile_out, "w") as f:
                json.dump(dict(microstep_ms_eager=tot_ms, classes=breakdown), f, indent=1)

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
                                   f"local batch {lb} x grad-accum {ga} x {world} GPU = global batch 32, Raven AdamW (bf16 m/v in pinned host memory, "
                                   "sharded 1/N per rank), clip 1.0, " + ("hipGraph replay" if a.graph else "eager 2-stream issue (dgrad chain || wgrad branch)"),
                       "global_batch": GLOBAL_BATCH, "parallelism": f"dp{world}"},
            "model_tflops_per_gpu": TRAIN_TFLOP_PER_SAMPLE * GLOBAL_BATCH / world * its,
            "mfma_roofline_frac_whole_step": TRAIN_TFLOP_PER_SAMPLE * GLOBAL_BATCH / world * its / PEAK_BF16_TFLOPS,
            "last_loss": loss_v, "last_grad_norm": gn_v,
            "roofline": roof, "cpu_baseline": cpu,
        }
        if a.rehearse_gloo:
            out["rehearsal"] = "mini UNet over gloo on one GPU: control-flow check only, the numbers are meaningless"
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()                    # rank 0 was still profiling: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
