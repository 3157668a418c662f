"""SURVEY 8f row f3 on the device: a run that is saved after step 1 (single-file model + training-state .pt) and
resumed into fresh objects must continue BITWISE like the uninterrupted run (every kernel is deterministic)."""
import os
import sys
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = "cuda:0"


def test_save_resume_continues_bitwise(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from safetensors.torch import save_file
    from aozora_sdxl_training_amd import checkpoint as C
    from aozora_sdxl_training_amd.unet import AozoraUNet
    from aozora_sdxl_training_amd.unet_spec import mini_config, param_table
    from aozora_sdxl_training_amd.train_step import TrainStep
    from aozora_sdxl_training_amd.optimizers import RavenAdamW
    from aozora_sdxl_training_amd.clip import clip_grad_norm_
    from aozora_sdxl_training_amd.schedule import TimestepSampler, CustomCurveLRScheduler, generate_noise
    cfg = mini_config()
    g = torch.Generator().manual_seed(11)
    names = [n for n, _ in param_table(cfg)]
    km = C.unet_key_mapping(names)
    base = {km[n]: ((torch.ones(shape) if n.endswith("weight") else torch.zeros(shape)) if "norm" in n
                    else torch.randn(*shape, generator=g) * 0.05).to(torch.bfloat16) for n, shape in param_table(cfg)}
    base["first_stage_model.decoder.conv_in.weight"] = torch.randn(4, 4, 3, 3, generator=g)
    base_path = tmp_path / "base.safetensors"
    save_file(base, str(base_path))

    B, h, w, GA, SEED = 2, 16, 16, 2, 42
    lat = torch.randn(B, 4, h, w, generator=g).bfloat16()
    ctx = torch.randn(B, 77, cfg.cross_attention_dim, generator=g).bfloat16()
    pooled = torch.randn(B, cfg.pooled_dim, generator=g).bfloat16()
    tid = torch.tensor([[128, 128, 0, 0, 128, 128]] * B, dtype=torch.bfloat16)
    run_cfg = types.SimpleNamespace(MAX_TRAIN_STEPS=2 * GA + 1, BATCH_SIZE=B, SEED=SEED, TIMESTEP_ALLOCATION=None)

    def make(model_path):
        unet = C.load_unet(model_path, DEV, cfg)                       # train.py:2606 load_unet_robust
        opt = RavenAdamW([{"params": list(unet.parameters()), "lr_scale": 1.0}], lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01,
                         eps=1e-8, debias_strength=0.3, momentum_dtype=torch.bfloat16)
        sched = CustomCurveLRScheduler(opt, [[0.0, 1e-4], [1.0, 2e-5]], run_cfg.MAX_TRAIN_STEPS)
        return unet, opt, sched, TimestepSampler(run_cfg), TrainStep(unet, mode="v_prediction", grad_accum=GA, use_graph=False)

    def one_optimizer_step(unet, opt, sched, sampler, step, micro):
        gen = torch.Generator()
        for _ in range(GA):
            micro += 1
            sched.step(micro)
            ts, _ = sampler.sample(B)
            noise = generate_noise(lat, gen, "cpu", step=micro, seed=SEED)
            step.micro_step(lat.to(DEV), noise.to(DEV), ts, ctx.to(DEV), pooled.to(DEV), tid.to(DEV))
        unet.expose_grads()
        raw = clip_grad_norm_(unet, 1.0).item()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return micro, raw

    # ---- uninterrupted: two optimizer steps
    u0, o0, s0, t0, st0 = make(base_path)
    micro, _ = one_optimizer_step(u0, o0, s0, t0, st0, 0)
    # save after step 1 (train.py:2513-2531)
    model_name, state_name = C.checkpoint_names("mini", 1)
    merged, missing = C.save_model(tmp_path / model_name, u0, base_path, torch.bfloat16)
    assert merged == len(names) and not missing
    C.save_training_state(tmp_path / state_name, 1, micro, o0, sampler_seed=SEED, sampler_epoch=1, timestep_sampler=t0)
    micro_end, raw_a = one_optimizer_step(u0, o0, s0, t0, st0, micro)
    torch.cuda.synchronize()
    final_a = u0.pflat.clone()

    # ---- resumed into fresh objects (train.py:2558-2573, 2684-2688)
    rs = C.load_training_state(tmp_path / state_name, grad_accum=GA)
    assert rs["optimizer_step"] == 1 and rs["micro_step"] == GA
    u1, o1, s1, t1, st1 = make(tmp_path / model_name)
    t1.load_state_dict(rs["timestep_sampler_state"])
    C.resume_optimizer(o1, rs["optimizer_state"], s1, rs["micro_step"])
    _, raw_b = one_optimizer_step(u1, o1, s1, t1, st1, rs["micro_step"])
    torch.cuda.synchronize()
    assert raw_a == raw_b
    assert torch.equal(u1.pflat, final_a)
    for n, p in list(u1.named_parameters())[::29]:
        assert o1.state[p]["step"] == 2 and torch.equal(o1.state[p]["exp_avg"], o0.state[u0._params[n]]["exp_avg"])
