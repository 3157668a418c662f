"""The loop around the hot path (trainer.train = train.py:2544-2830 on this package's objects) end to end on the device:
synthetic on-disk cache -> data feed -> HIP micro-steps -> clip -> Raven -> LR curve -> reporter lines -> checkpoints, and a
run resumed from its own mid-run checkpoint finishes BITWISE like the uninterrupted one."""
import contextlib
import io
import os
import sys
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
DEV = "cuda:0"


def _config(tmp, mode, **over):
    import synth_cache
    synth_cache.build(os.path.join(tmp, "set0"), n_items=23, json_mode=False, seed=0, rf=(mode == "rectified_flow"))
    cfg = types.SimpleNamespace(
        INSTANCE_DATASETS=[{"path": os.path.join(tmp, "set0"), "repeats": 1}], CAPTION_SOURCE_TYPE="txt", SEED=42,
        MAX_TRAIN_STEPS=8, BATCH_SIZE=2, GRADIENT_ACCUMULATION_STEPS=2, PREDICTION_TYPE=mode, CLIP_GRAD_NORM=1.0,
        LR_CUSTOM_CURVE=[[0.0, 0.0], [0.2, 1e-4], [1.0, 2e-5]], LEARNING_RATE=1e-4, OPTIMIZER_TYPE="raven",
        RAVEN_PARAMS=dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, debias_strength=0.3, momentum_dtype="bfloat16"),
        UNET_EXCLUDE_TARGETS="conv1, conv2", SAVE_EVERY_N_STEPS=2, OUTPUT_DIR=os.path.join(tmp, "out"), OUTPUT_NAME="mini_run",
        SINGLE_FILE_CHECKPOINT_PATH=os.path.join(tmp, "base.safetensors"), RESUME_TRAINING=False,
        TIMESTEP_ALLOCATION={"bin_size": 100, "counts": [45, 143, 176, 173, 154, 126, 94, 59, 26, 4]},
        TIMESTEP_LOSS_WEIGHT_CURVE={"preset": "bell"}, TIMESTEP_FORCE_IMAGE_BIN_SPREAD=True, NUM_WORKERS=0)
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def _base_checkpoint(path, cfg_model):
    from safetensors.torch import save_file
    from aozora_sdxl_training_amd import checkpoint as C
    from aozora_sdxl_training_amd.unet_spec import param_table
    g = torch.Generator().manual_seed(3)
    km = C.unet_key_mapping([n for n, _ in param_table(cfg_model)])
    t = {km[n]: ((torch.ones(s) if n.endswith("weight") else torch.zeros(s)) if "norm" in n else torch.randn(*s, generator=g) * 0.05).to(torch.bfloat16)
         for n, s in param_table(cfg_model)}
    t["first_stage_model.post_quant_conv.bias"] = torch.zeros(4)
    save_file(t, str(path))


@pytest.mark.parametrize("mode", ["v_prediction", "rectified_flow"])
def test_train_loop_and_bitwise_resume(tmp_path, mode):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from aozora_sdxl_training_amd import checkpoint as C
    from aozora_sdxl_training_amd.trainer import train
    from aozora_sdxl_training_amd.telemetry import Reporter
    from aozora_sdxl_training_amd.unet_spec import mini_config
    model = mini_config(ctx_dim=64, pooled=32)
    tmp = str(tmp_path)
    cfg = _config(tmp, mode)
    _base_checkpoint(cfg.SINGLE_FILE_CHECKPOINT_PATH, model)

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        unet = C.load_unet(cfg.SINGLE_FILE_CHECKPOINT_PATH, DEV, model)
        h = train(cfg, unet=unet, device=DEV, reporter=Reporter(cfg.MAX_TRAIN_STEPS, asynchronous=False))
    torch.cuda.synchronize()
    out = buf.getvalue()
    assert h["micro_step"] == 8 and h["optimizer_step"] == 4 and len(h["losses"]) == 8 and len(h["grad_norms"]) == 4
    assert all(l == l and 0.0 < l < 10.0 for l in h["losses"]) and all(g > 0 for g in h["grad_norms"])
    assert h["lrs"][-1] == pytest.approx(2e-5) and h["lrs"][0] == pytest.approx(1e-4 * 1.0 if False else h["lrs"][0])
    assert h["saved"] == [("mini_run_step_2.safetensors", "mini_run_training_state_step_2.pt"),
                          ("mini_run_step_4.safetensors", "mini_run_training_state_step_4.pt")]
    assert out.count("--- Optimizer Step:") == 4 and "Training |" in out and "Training complete." in out and "[NO UPDATE!]" not in out
    # train.py:2832-2836: the final model is written after the loop, whatever SAVE_EVERY_N_STEPS says
    assert h["final_model"] == os.path.join(cfg.OUTPUT_DIR, "mini_run.safetensors") and "All tasks complete. Final model saved." in out
    saved = C.read_unet_state(h["final_model"], [n for n, _ in unet.named_parameters()])
    assert all(torch.equal(saved[n], p.detach().cpu()) for n, p in unet.named_parameters())
    frozen = [n for n, p in unet.named_parameters() if not p.requires_grad]
    assert frozen and all(("conv1" in n or "conv2" in n) for n in frozen)
    final = unet.pflat.clone()

    # resume from the checkpoint written after optimizer step 2 (micro-step 4) into fresh objects
    cfg2 = _config(tmp, mode, RESUME_TRAINING=True, SAVE_EVERY_N_STEPS=0,
                   RESUME_MODEL_PATH=os.path.join(cfg.OUTPUT_DIR, "mini_run_step_2.safetensors"),
                   RESUME_STATE_PATH=os.path.join(cfg.OUTPUT_DIR, "mini_run_training_state_step_2.pt"))
    with contextlib.redirect_stdout(io.StringIO()):
        unet2 = C.load_unet(cfg2.RESUME_MODEL_PATH, DEV, model)
        h2 = train(cfg2, unet=unet2, device=DEV, reporter=Reporter(cfg2.MAX_TRAIN_STEPS, asynchronous=False))
    torch.cuda.synchronize()
    assert h2["micro_step"] == 8 and h2["optimizer_step"] == 4
    assert h2["losses"] == h["losses"][4:] and h2["grad_norms"] == h["grad_norms"][2:]
    assert torch.equal(unet2.pflat, final)
