"""aozora_sdxl_training_amd -- MI355X-native SDXL UNet training step (hand-written HIP behind the
reference's train.py loop / Raven-Titan optimizer API).  See DESIGN.md."""
import os as _os

# one hardware queue per stream the step uses (see streams.py); only effective if set before the HIP runtime initialises
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
__version__ = "0.1.0"
