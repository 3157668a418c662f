"""aozora_sdxl_training_amd -- MI355X-native SDXL UNet training step (hand-written HIP behind the
reference's train.py loop / Raven-Titan optimizer API).  See DESIGN.md."""
__version__ = "0.1.0"
