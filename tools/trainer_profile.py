"""Where the trainer loop's HOST time goes: cProfile of trainer.train on the bench's workload (full-size UNet, B = 4, GA = 8; the first
optimizer step -- pools, launch tape -- is included, so read the per-call columns).  python tools/trainer_profile.py [optimizer steps]"""
import cProfile, io, os, pstats, sys, tempfile, types, torch
sys.path.insert(0, '.')
import bench
from aozora_sdxl_training_amd.unet import AozoraUNet
from aozora_sdxl_training_amd.unet_spec import SDXL_BASE
from aozora_sdxl_training_amd.trainer import train
from safetensors.torch import save_file
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device('cuda', 0)
unet = AozoraUNet(SDXL_BASE, dev); bench.init_weights_on_device(unet)
tmp = tempfile.mkdtemp()
bench.write_synthetic_cache(os.path.join(tmp, 'set0'), 8, bench.LATENT, SDXL_BASE.cross_attention_dim, SDXL_BASE.pooled_dim)
save_file({'placeholder.weight': torch.zeros(1)}, os.path.join(tmp, 'base.safetensors'))
cfg = types.SimpleNamespace(
    INSTANCE_DATASETS=[{'path': os.path.join(tmp, 'set0'), 'repeats': 1}], CAPTION_SOURCE_TYPE='txt', SEED=42, MAX_TRAIN_STEPS=8 * iters,
    BATCH_SIZE=4, GRADIENT_ACCUMULATION_STEPS=8, PREDICTION_TYPE='epsilon', CLIP_GRAD_NORM=1.0,
    LR_CUSTOM_CURVE=[[0.0, 0.0], [0.05, 8.0e-7], [0.85, 8.0e-7], [1.0, 1.0e-7]], LEARNING_RATE=8e-7, OPTIMIZER_TYPE='raven',
    RAVEN_PARAMS=dict(betas=[0.9, 0.999], eps=1e-8, weight_decay=0.01, debias_strength=0.3, momentum_dtype='bfloat16'),
    UNET_EXCLUDE_TARGETS=[], SAVE_EVERY_N_STEPS=0, OUTPUT_DIR=os.path.join(tmp, 'out'), OUTPUT_NAME='prof',
    SINGLE_FILE_CHECKPOINT_PATH=os.path.join(tmp, 'base.safetensors'), RESUME_TRAINING=False, TIMESTEP_ALLOCATION=None,
    TIMESTEP_LOSS_WEIGHT_CURVE=None, TIMESTEP_FORCE_IMAGE_BIN_SPREAD=False, NUM_WORKERS=0)
class Quiet:
    def log_step(self, *a, **k): pass
    def log_message(self, *a, **k): pass
    def shutdown(self): pass
pr = cProfile.Profile(); pr.enable()
train(cfg, unet=unet, device=str(dev), reporter=Quiet())
pr.disable()
out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats('tottime').print_stats(22)
print(out.getvalue()[:6000])
