"""north_star's parity bar at FULL scale: the real SDXL-base UNet (2.567 B parameters, diffusers names) on the HIP path against
the CPU oracle on identical bf16-rounded weights, latents, noise and timesteps -- per-step loss and global gradient norm.

The bar (asserted): 1e-3 relative, for BOTH observables, against the oracle run in the reference's own arithmetic: bf16
parameters under bf16 autocast, scheduler coefficients rounded to the latents' bf16 (train.py:273 -- the reference has no fp32
mode).  The distance to the all-fp32 oracle is measured, printed and written to gpurun_out/ beside it; it is
gated only by "no worse than 2x the reference dataflow's own distance from fp32" (bf16 storage between layers moves the
gradient norm by 1-2e-3 whoever computes it), with no fixed cap.

Cases (BASELINE.json configs, at the sizes the oracle finishes in tens of seconds on the host):
  * eps, 256x256, B=1                                   the smallest square bucket
  * v-prediction, 224x160 ragged, B=2, 154 context tokens (chunked captions), timesteps 23 / 871
  * cfg1: eps, 512x512, B=1, INCLUDING the clip + Raven step (parameter update against the oracle's)
  * cfg3: v-prediction + logit-normal tickets, 512x512, B=2, grad-accum 2 (tickets / noise drawn exactly as the trainer does)
  * cfg4: rectified flow, grad-accum 2, the bucket SWITCHES inside one TrainStep: micro-step 1 on the 768x768 bucket, micro-step 2
    on the 896x896 one (T = 3136 / 784: ragged against every tile size) -- one activation pool and launch tape per bucket
    (jitter from the LCG-seeded generator, train.py:2743-2752; bucket-homogeneous batches: train.py:486-500, 2645-2655)
  * cfg5: freeze keywords "mid_block, up_blocks.3" + Titan, v-prediction, 512x512, grad-accum 2: single-GPU TitanAdamW (host fp32
    gradient buffer, host clip, step) AND dist.ShardedTitan at world 1 over RCCL, both against the oracle's Titan chain
  * cfg2's resolution: eps, 1024x1024, B=1 (the bench runs B=4 of these: T = 4096 / 1024 attention, 128^2 convolutions) -- 60-70 s of oracle
(cfg2's batch itself -- B=4 at 1024x1024 -- is covered by the size-independent properties below.)"""
import json
import math
import os
import sys
import time

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = "cuda:0"
LOGIT_NORMAL = {"bin_size": 100, "counts": [45, 143, 176, 173, 154, 126, 94, 59, 26, 4]}      # SURVEY 8c F5 (mu -0.5, sigma 1)


def _gn(grads):
    return float(torch.sqrt(sum(g_.double().pow(2).sum() for g_ in grads.values())))


def _micro_inputs(mode, B, h, w, ntok, ga, tsteps, seed=42):
    """Per micro-step (1-based, as train.py:2713): latents / context from a fixed generator; timesteps given or drawn from
    the ticket pool; noise and rectified-flow jitter exactly as trainer.train draws them (schedule.generate_noise,
    schedule.seeded_torch_generator: bit-exact against the reference's, tests/test_host_golden.py)."""
    from aozora_sdxl_training_amd.schedule import build_timestep_ticket_pool, generate_noise, seeded_torch_generator
    g = torch.Generator().manual_seed(5 if seed == 42 else 1000 + seed)      # (seed 42: the latents / context of rounds 1-4)
    pool = None
    if tsteps is None:
        pool, _ = build_timestep_ticket_pool(LOGIT_NORMAL, ga * B, 1000, seed, False)
    out, ngen = [], torch.Generator()
    for ms in range(1, ga + 1):
        lat = torch.randn(B, 4, h, w, generator=g).bfloat16()
        ctx = torch.randn(B, ntok, 2048, generator=g).bfloat16()
        pooled = torch.randn(B, 1280, generator=g).bfloat16()
        tid = torch.tensor([[h * 8, w * 8, 0, 0, h * 8, w * 8]] * B, dtype=torch.bfloat16)
        ts = torch.tensor(tsteps if pool is None else pool[(ms - 1) * B: ms * B])
        noise = generate_noise(lat, ngen, "cpu", step=ms, seed=seed)
        jit = None
        if mode == "rectified_flow":
            jit = torch.rand(ts.shape, dtype=torch.float32, generator=seeded_torch_generator("cpu", seed, ms, 0x5D1))
        out.append((lat, noise, ts, ctx, pooled, tid, jit))
    return out


_SHARED = {}


@pytest.fixture(scope="module", autouse=True)
def _release_full_size_model():
    """The shared full-size UNet (weights, gradient / W^T buffers, one activation pool per bucket: ~100 GB of HBM) and the
    10 GB of host parameters go away with the module instead of staying cached under whatever runs next."""
    yield
    import gc
    _SHARED.clear()
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


def _shared():
    """Seed-generated SDXL-base weights (bf16-rounded, fp32 storage: 10 GB of host memory) and ONE AozoraUNet, built once for
    all cases of this module (each case reloads the weights: the Raven case changes them)."""
    if not _SHARED:
        from oracle.unet_ref import SDXL_BASE as OCFG, init_params
        from aozora_sdxl_training_amd.unet import AozoraUNet
        from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
        torch.set_num_threads(min(len(os.sched_getaffinity(0)), 64))
        _SHARED["params"] = {k: v.bfloat16().float() for k, v in init_params(OCFG, seed=1234).items()}
        _SHARED["unet"] = AozoraUNet(SDXL_BASE, DEV)
    return _SHARED["params"], _SHARED["unet"]


CASES = [
    # id, mode, B, h, w, ctx tokens, grad-accum, timesteps (None: logit-normal tickets), fp32 oracle too, Raven step
    # (round 1's 256 px epsilon case is superseded by cfg1: same mode, four times the pixels, fp32 yardstick + Raven step)
    ("vpred_ragged", "v_prediction", 2, 20, 28, 154, 1, [23, 871], False, False),
    ("cfg1_eps512_raven", "epsilon", 1, 64, 64, 77, 1, [417], True, True),
    # the bench's own resolution (T = 4096 / 1024, 128^2 convs): three (timestep, seed) samples -- other latents, context, noise --
    # and the all-fp32 yardstick on the first (round 5: one sample at 9.06e-4 of the 1e-3 gate says little about the margin)
    ("cfg2_eps1024_b1", "epsilon", 1, 128, 128, 77, 1, [417], "ref_inputs", False),      # (the everything-fp32 leg ran in round 5: HIP 1.83e-3, reference dataflow 1.23e-3)
    ("cfg2_eps1024_b1_t23_seed7", "epsilon", 1, 128, 128, 77, 1, [23], False, False, 7),
    ("cfg2_eps1024_b1_t871_seed11", "epsilon", 1, 128, 128, 77, 1, [871], False, False, 11),
    ("cfg3_vpred512_tickets_ga2", "v_prediction", 2, 64, 64, 77, 2, None, False, False),
    ("cfg4_rf_768_then_896_ga2", "rectified_flow", 1, [96, 112], [96, 112], 77, 2, [105, 640], False, False),
]


@pytest.mark.parametrize("cid,mode,B,h,w,ntok,ga,tsteps,with_fp32,raven,seed", [c if len(c) == 11 else c + (42,) for c in CASES], ids=[c[0] for c in CASES])
def test_full_sdxl_unet_step_matches_oracle(cid, mode, B, h, w, ntok, ga, tsteps, with_fp32, raven, seed):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle.unet_ref import SDXL_BASE as OCFG, init_params
    from oracle.step_ref import RefTrainer
    from aozora_sdxl_training_amd.train_step import TrainStep
    from aozora_sdxl_training_amd.clip import clip_grad_norm_
    from aozora_sdxl_training_amd.optimizers import RavenAdamW
    params, unet = _shared()
    if tsteps is not None and ga > 1:
        steps = [[t] * B for t in tsteps]                 # one timestep list per micro-step
    else:
        steps = [tsteps] * ga
    micro = []
    for i in range(ga):       # h / w may be lists: one resolution bucket per micro-step (cfg4: the bucket changes between batches)
        hi, wi = (h[i], w[i]) if isinstance(h, list) else (h, w)
        micro.append(_micro_inputs(mode, B, hi, wi, ntok, ga, steps[i], seed=seed)[i])
    LR = 1e-4           # large enough that one AdamW step moves bf16 parameters by whole ulps (the default 8e-7 rounds away)
    rep = dict(case=cid, timesteps=[m[2].tolist() for m in micro], seed=seed)
    l_ref = gn_ref = l_ri = gn_ri = None
    if with_fp32:
        # two fp32 yardsticks: (a) everything fp32, the scheduler coefficients included; (b) fp32 UNet / loss / backward on the
        # reference's own inputs (coefficients cast to bf16 before the square root, x_t rounded to bf16: oracle/step_ref.py
        # RefTrainer(ref_inputs=True)).  (a) also measures a coherent scale on x_t that belongs to the reference's dataflow
        # (t = 417: +2.6e-3), which every GroupNorm's rstd hands on to the gradients; (b) measures arithmetic only.
        t0 = time.time()
        if with_fp32 != "ref_inputs":
            ref = RefTrainer(OCFG, params, mode=mode, bf16=False, ga=ga, clip=1.0)
            l_ref = [ref.micro_step(*m[:6], jitter=m[6]) for m in micro]
            gn_ref = _gn(ref.grads())
            del ref
        ref = RefTrainer(OCFG, params, mode=mode, bf16=False, ga=ga, clip=1.0, ref_inputs=True)
        l_ri = [ref.micro_step(*m[:6], jitter=m[6]) for m in micro]
        gn_ri = _gn(ref.grads())
        del ref
        rep["oracle_fp32_s"] = time.time() - t0
    t0 = time.time()
    ref16 = RefTrainer(OCFG, params, mode=mode, bf16=True, ga=ga, clip=1.0, lr=LR)
    l_16 = [ref16.micro_step(*m[:6], jitter=m[6]) for m in micro]
    gn_16 = _gn({k: v.float() for k, v in ref16.grads().items()})
    rep["oracle_bf16_s"] = time.time() - t0
    unet.load_state_dict(params)
    step = TrainStep(unet, mode=mode, grad_accum=ga, use_graph=False)
    unet.zero_grad()
    l_hip = [step.micro_step(m[0].to(DEV), m[1].to(DEV), m[2], m[3].to(DEV), m[4].to(DEV), m[5].to(DEV), m[6]).item() for m in micro]
    unet.expose_grads()
    before = unet.pflat.clone() if raven else None
    gn_hip = clip_grad_norm_(list(unet.parameters()), 1.0).item()
    rel = lambda a, b: abs(a - b) / abs(b)
    rep.update(loss_hip=l_hip, loss_bf16_oracle=l_16, loss_fp32=l_ref, gn_hip=gn_hip, gn_bf16_oracle=gn_16, gn_fp32=gn_ref,
               loss_rel_vs_bf16_oracle=[rel(a, b) for a, b in zip(l_hip, l_16)], gn_rel_vs_bf16_oracle=rel(gn_hip, gn_16))
    if with_fp32 and l_ref is not None:
        rep.update(loss_rel_vs_fp32=[rel(a, b) for a, b in zip(l_hip, l_ref)], gn_rel_vs_fp32=rel(gn_hip, gn_ref),
                   bf16_oracle_loss_rel_vs_fp32=[rel(a, b) for a, b in zip(l_16, l_ref)], bf16_oracle_gn_rel_vs_fp32=rel(gn_16, gn_ref))
    if with_fp32:
        rep.update(loss_fp32_ref_inputs=l_ri, gn_fp32_ref_inputs=gn_ri,
                   loss_rel_vs_fp32_ref_inputs=[rel(a, b) for a, b in zip(l_hip, l_ri)], gn_rel_vs_fp32_ref_inputs=rel(gn_hip, gn_ri),
                   bf16_oracle_loss_rel_vs_fp32_ref_inputs=[rel(a, b) for a, b in zip(l_16, l_ri)],
                   bf16_oracle_gn_rel_vs_fp32_ref_inputs=rel(gn_16, gn_ri))
    if raven:
        opt = RavenAdamW([{"params": list(unet.parameters()), "lr_scale": 1.0}], lr=LR, betas=(0.9, 0.999), weight_decay=0.01, eps=1e-8,
                         debias_strength=0.3, momentum_dtype=torch.bfloat16)
        opt.step()
        torch.cuda.synchronize()
        big16 = {}
        for k, v in ref16.grads().items():                # elements whose gradient magnitude is in the upper half of their tensor
            if v.numel() >= 65536:
                a = v.float().abs()
                big16[k] = a >= a.flatten().kthvalue(a.numel() // 2).values
        raw16 = ref16.optimizer_step()                    # clip 1.0 + AdamW on the oracle's bf16 parameters
        rep["raw_norm_bf16_oracle_as_torch_clips"] = raw16
        # parameter update, element by element: step-1 AdamW moves every element by ~3.6 lr in the direction of -sign(g), so
        # elements whose gradient is below the bf16 noise of the two dataflows may legitimately differ in sign; the rest must agree
        agree_n = agree_d = 0
        sq_d = sq_r = 0.0
        for name, p in unet.named_parameters():
            if p.numel() < 65536:
                continue
            o, st, shape = unet._slots[name]
            n = math.prod(st)
            d_h = (unet.pflat[o:o + n].float() - before[o:o + n].float()).view(st)
            if len(st) == 4:
                d_h = d_h.permute(0, 3, 1, 2)[:, :shape[1]]
            d_h = d_h.cpu()
            d_r = ref16.params[name].detach().float() - params[name]
            big = big16[name]
            agree_n += int(((torch.sign(d_h) == torch.sign(d_r)) & big).sum())
            agree_d += int(big.sum())
            sq_d += float((d_h - d_r).double().pow(2).sum())
            sq_r += float(d_r.double().pow(2).sum())
        rep.update(update_sign_agreement_upper_half=agree_n / agree_d, update_rel_l2=math.sqrt(sq_d / sq_r))
    del ref16
    print("full-size parity:", json.dumps(rep))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"fullsize_parity_{cid}.json"), "w") as f:
        json.dump(rep, f, indent=1)
    for a, b in zip(l_hip, l_16):
        assert rel(a, b) <= 1e-3, ("loss vs bf16-autocast oracle", rep)
    assert rel(gn_hip, gn_16) <= 1e-3, ("grad-norm vs bf16-autocast oracle", rep)
    if with_fp32:
        # (b) fp32 arithmetic on the reference's own inputs: north_star's "within 1e-3 rel fp32", as a hard gate (measured round 5:
        #     gradient norm 2.4e-4 at 512^2, 4.7e-4 at 1024^2; the reference dataflow itself: 2.4e-4 / 1.4e-4).
        for a_, b_ in zip(l_hip, l_ri):
            assert rel(a_, b_) <= 1e-3, ("loss vs fp32 on the reference's inputs", rep)
        assert rel(gn_hip, gn_ri) <= 1e-3, ("grad-norm vs fp32 on the reference's inputs", rep)
        # (a) everything fp32, scheduler coefficients included: this distance is mostly the reference dataflow's OWN (its bf16 coefficients
        #     scale x_t coherently -- t = 417: +2.6e-3 -- and every GroupNorm's rstd hands that to the gradients; tools/act_noise.py,
        #     profiles/r05_parity_localisation.md): reference 1.03e-3 / 1.23e-3, HIP 1.03e-3 / 1.83e-3 at 512^2 / 1024^2.  Gate: 1.6 x the
        #     reference dataflow's own distance.
        if l_ref is not None:
            for a_, b_, c_ in zip(l_hip, l_ref, l_16):
                assert rel(a_, b_) <= max(1e-3, 1.6 * rel(c_, b_)), ("loss vs fp32", rep)
            assert rel(gn_hip, gn_ref) <= max(1e-3, 1.6 * rel(gn_16, gn_ref)), ("grad-norm vs fp32", rep)
    if raven:       # measured: 0.9990 sign agreement, 0.060 relative L2 (profiles / DESIGN.md section 2)
        assert rep["update_sign_agreement_upper_half"] >= 0.995 and rep["update_rel_l2"] <= 0.10, rep
    if isinstance(h, list):
        assert len(step._buckets) == len(set(zip(h, w))), "one activation pool / launch tape per resolution bucket"


def _update_agreement(unet, before, after_ref, params, big, names):
    """Parameter update HIP vs oracle over `names`: sign agreement on the elements whose gradient magnitude is in the upper half of
    their tensor (`big`), and relative L2 of the update difference."""
    agree_n = agree_d = 0
    sq_d = sq_r = 0.0
    for name in names:
        o, st, shape = unet._slots[name]
        n = math.prod(st)
        d_h = (unet.pflat[o:o + n].float() - before[o:o + n].float()).view(st)
        if len(st) == 4:
            d_h = d_h.permute(0, 3, 1, 2)[:, :shape[1]]
        d_h = d_h.cpu()
        d_r = after_ref[name].float() - params[name]
        agree_n += int(((torch.sign(d_h) == torch.sign(d_r)) & big[name]).sum())
        agree_d += int(big[name].sum())
        sq_d += float((d_h - d_r).double().pow(2).sum())
        sq_r += float(d_r.double().pow(2).sum())
    return agree_n / max(agree_d, 1), math.sqrt(sq_d / max(sq_r, 1e-300))


def test_cfg5_freeze_keywords_and_titan_at_full_size():
    """BASELINE configs[4] on the real SDXL-base UNet: UNET_EXCLUDE_TARGETS = "mid_block, up_blocks.3" (train.py:2664-2667: 413 M
    parameters frozen, up_blocks.3 matches nothing), v-prediction with logit-normal tickets, 512x512, grad-accum 2, Titan:
      (A) single-GPU TitanAdamW -- offload_flat after every micro-step into the pinned fp32 host buffer (titan.py:119-131),
          clip_grad_norm on the host gradients (162-184), step (230-296);
      (B) dist.ShardedTitan at world 1 over RCCL (force_exchange: fp32 reduce-scatter, scalar all-reduce, all-gather issued as with N ranks);
    both against the oracle's Titan chain (titan_accumulate / titan_clip / adamw_debiased_step on the bf16-autocast oracle's
    gradients): per-micro-step loss and the raw fp32 norm within 1e-3, the update direction / size, frozen ranges untouched."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import socket
    import torch.distributed as dist
    from oracle.unet_ref import SDXL_BASE as OCFG
    from oracle.step_ref import RefTrainer, titan_accumulate, titan_clip, adamw_debiased_step
    from aozora_sdxl_training_amd.train_step import TrainStep
    from aozora_sdxl_training_amd.optimizers import TitanAdamW
    from aozora_sdxl_training_amd.dist import ShardedTitan
    from aozora_sdxl_training_amd.schedule import trainable_mask
    params, unet = _shared()
    names = [n for n, _ in unet.named_parameters()]
    mask = trainable_mask(names, ["mid_block", "up_blocks.3"])
    frozen = tuple(n for n, m in zip(names, mask) if not m)
    assert sum(math.prod(unet._slots[n][2]) for n in frozen) == 413_117_440 and all(n.startswith("mid_block.") for n in frozen)
    mode, B, hw, GA, LR, CLIP = "v_prediction", 1, 64, 2, 1e-4, 1.0
    HP = dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, debias_strength=0.3, momentum_dtype=torch.bfloat16)
    micro = _micro_inputs(mode, B, hw, hw, 77, GA, None, seed=42)
    rel = lambda a, b: abs(a - b) / abs(b)
    # ---- oracle: bf16-autocast dataflow (train.py:273), Titan arithmetic on its per-micro-step bf16 gradients ----
    t0 = time.time()
    ref = RefTrainer(OCFG, params, mode=mode, bf16=True, ga=GA, clip=CLIP, lr=LR, frozen=frozen)
    acc, l_ref = {}, []
    for m in micro:
        l_ref.append(ref.micro_step(*m[:6], jitter=m[6]))
        for n, gr in ref.grads().items():
            acc[n] = titan_accumulate(acc.get(n), gr)
        for p_ in ref.params.values():
            p_.grad = None
    assert not (set(acc) & set(frozen)) and len(acc) == len(names) - len(frozen)
    big = {}
    for n, g_ in acc.items():
        if g_.numel() >= 65536:
            a = g_.abs()
            big[n] = a >= a.flatten().kthvalue(a.numel() // 2).values
    raw_ref = float(titan_clip(list(acc.values()), CLIP))
    after_ref = {}
    with torch.no_grad():
        for n, p_ in ref.params.items():
            if n not in acc:
                continue
            m_, v_ = torch.zeros_like(p_, dtype=torch.bfloat16), torch.zeros_like(p_, dtype=torch.bfloat16)
            adamw_debiased_step(p_, acc[n], m_, v_, 1, LR, 0.9, 0.999, 1e-8, 0.01, 0.3)
            if n in big:
                after_ref[n] = p_.detach().float().clone()
    del ref, acc
    rep = dict(case="cfg5_freeze_titan", oracle_s=time.time() - t0, loss_bf16_oracle=l_ref, raw_norm_oracle=raw_ref)

    def set_mask():
        unet.load_state_dict(params)
        for (n, p), m in zip(unet.named_parameters(), mask):
            p.requires_grad = m
    try:
        # ---- (A) single-GPU TitanAdamW ----
        set_mask()
        tp = [p for p in unet.parameters() if p.requires_grad]
        opt = TitanAdamW([{"params": tp, "lr_scale": 1.0}], lr=LR, **HP)
        step = TrainStep(unet, mode=mode, grad_accum=GA, use_graph=False)
        unet.zero_grad()
        l_a = []
        for m in micro:
            l_a.append(step.micro_step(m[0].to(DEV), m[1].to(DEV), m[2], m[3].to(DEV), m[4].to(DEV), m[5].to(DEV), m[6]).item())
            opt.offload_flat(unet)
        before = unet.pflat.clone()
        raw_a = float(opt.clip_grad_norm(CLIP))
        opt.step()
        opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        after_a = unet.pflat.clone()
        agree_a, l2_a = _update_agreement(unet, before, after_ref, params, big, list(big))
        lo = min(unet._slots[n][0] for n in frozen); hi = max(unet._slots[n][0] + math.prod(unet._slots[n][1]) for n in frozen)
        frozen_ok_a = bool(torch.equal(after_a[lo:hi], before[lo:hi]))
        opt.close()
        del opt
        rep.update(loss_titan=l_a, raw_norm_titan=raw_a, titan_sign_agreement=agree_a, titan_update_rel_l2=l2_a, titan_frozen_untouched=frozen_ok_a)
        # ---- (B) ShardedTitan, world 1, RCCL ----
        sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device(DEV))
        try:
            set_mask()
            opt2 = ShardedTitan(unet, lr=LR, clip_grad_norm=CLIP, force_exchange=True, **HP)
            assert opt2.exchange and opt2.overlap
            step2 = TrainStep(unet, mode=mode, grad_accum=GA, world_size=1, use_graph=False)
            opt2.zero_grad()
            l_b = []
            hooked = []
            for i, m in enumerate(micro):       # the last micro-step hands regions 2 and 1 to the fp32 exchange from inside its backward
                hook = (lambda k: (hooked.append(k), opt2.reduce_tail(k))) if i == len(micro) - 1 else None
                l_b.append(step2.micro_step(m[0].to(DEV), m[1].to(DEV), m[2], m[3].to(DEV), m[4].to(DEV), m[5].to(DEV), m[6], after_tail=hook).item())
                opt2.accumulate()
            assert sorted(set(hooked)) == [1, 2], hooked
            raw_b = opt2.step().item()
            unet.wait_tail_params()
            torch.cuda.synchronize()
            after_b = unet.pflat.clone()
            agree_b, l2_b = _update_agreement(unet, before, after_ref, params, big, list(big))
            d_a, d_b = after_a.float() - before.float(), after_b.float() - before.float()
            rep.update(loss_sharded=l_b, raw_norm_sharded=raw_b, sharded_sign_agreement=agree_b, sharded_update_rel_l2=l2_b,
                       sharded_vs_single_update_rel_l2=float((d_a - d_b).norm() / d_a.norm()),
                       sharded_frozen_untouched=bool(torch.equal(after_b[lo:hi], before[lo:hi])))
            opt2.synchronize_state()
            del opt2
        finally:
            dist.destroy_process_group()
    finally:
        for p in unet.parameters():
            p.requires_grad = True
    print("full-size parity:", json.dumps(rep))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "fullsize_parity_cfg5_freeze_titan.json"), "w") as f:
        json.dump(rep, f, indent=1)
    for a, b, c in zip(l_a, l_b, l_ref):
        assert rel(a, c) <= 1e-3 and a == b, ("loss vs bf16-autocast oracle / single vs sharded", rep)
    assert rel(raw_a, raw_ref) <= 1e-3 and rel(raw_b, raw_ref) <= 1e-3, ("raw fp32 Titan norm", rep)
    assert rep["titan_frozen_untouched"] and rep["sharded_frozen_untouched"], rep
    assert agree_a >= 0.995 and l2_a <= 0.10 and agree_b >= 0.995 and l2_b <= 0.10, rep
    assert rep["sharded_vs_single_update_rel_l2"] <= 0.02, rep


def test_full_size_properties_at_benchmark_shape():
    """BASELINE configs[1] shape (SDXL-base UNet, B=4, 1024x1024 => latents 4x128x128), far beyond what the CPU oracle can
    check directly; size-independent properties instead:
      * determinism: two runs of the same micro-step give bit-identical loss and gradient buffer (ordered reductions only);
      * linearity: scaling the per-sample loss weights by 2 (exact in binary floating point) scales the loss and EVERY
        gradient element by exactly 2 -- through all 3 300 launches of forward + backward;
      * the launch-tape replay equals the eager issue bit for bit at this size;
      * freezing a block leaves the other gradients untouched and its own at zero."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import bench
    from aozora_sdxl_training_amd.unet import AozoraUNet
    from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
    from aozora_sdxl_training_amd.train_step import TrainStep
    dev = torch.device("cuda", 0)
    unet = AozoraUNet(SDXL_BASE, dev)
    bench.init_weights_on_device(unet)
    batch = bench.synthetic_batch(0, 0, 0, 4, dev)
    step = TrainStep(unet, mode="epsilon", grad_accum=1, use_graph=False)

    def run(scale=1.0):
        unet.zero_grad()
        l = step.micro_step(*batch, weight_scale=scale)
        torch.cuda.synchronize()
        return float(l.item()), unet.gflat.clone()

    l0, g0 = run()          # eager (allocates the pools)
    l1, g1 = run()          # eager, recorded
    l2, g2 = run()          # tape replay
    assert l0 == l1 == l2 and torch.equal(g0, g1) and torch.equal(g1, g2)
    assert 0.5 < l0 < 2.0 and float(g0.float().abs().max()) > 0
    ls, gs = run(2.0)
    assert ls == 2.0 * l0
    assert torch.equal(gs.float(), 2.0 * g0.float())
    # freeze the mid block: its gradient range stays zero, everything else is unchanged
    mid = [(n, p) for n, p in unet.named_parameters() if n.startswith("mid_block.")]
    for _, p in mid:
        p.requires_grad = False
    lf, gf = run()
    assert lf == l0
    slots = unet._slots
    import math
    lo = min(slots[n][0] for n, _ in mid); hi = max(slots[n][0] + math.prod(slots[n][1]) for n, _ in mid)
    assert float(gf[lo:hi].float().abs().max()) == 0.0
    # (diffusers order puts mid_block in one contiguous range of the flat buffer)
    assert torch.equal(gf[:lo], g0[:lo]) and torch.equal(gf[hi:], g0[hi:])
    for _, p in mid:
        p.requires_grad = True
