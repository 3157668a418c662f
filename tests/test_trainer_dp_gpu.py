"""trainer.train under data parallel: 2 ranks (sharing the test box's one GPU, gloo) over the same synthetic cache and config
must reproduce the single-process run at the same GLOBAL batch: per-micro-step mean loss and per-step global grad norm
(bf16 tolerance: 2e-3 / 1e-2), rank 0 writes the checkpoint, every rank its optimizer shard, and a 2-rank resume from those
files continues bitwise."""
import contextlib
import io
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
DEV = "cuda:0"


def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run(cfg, model):
    from aozora_sdxl_training_amd import checkpoint as C
    from aozora_sdxl_training_amd.trainer import train
    from aozora_sdxl_training_amd.telemetry import Reporter
    with contextlib.redirect_stdout(io.StringIO()):
        path = cfg.RESUME_MODEL_PATH if cfg.RESUME_TRAINING else cfg.SINGLE_FILE_CHECKPOINT_PATH
        unet = C.load_unet(path, DEV, model)
        h = train(cfg, unet=unet, device=DEV, reporter=Reporter(cfg.MAX_TRAIN_STEPS, asynchronous=False))
    torch.cuda.synchronize()
    return h, unet


def _worker(rank, world, port, tmp, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import test_trainer_gpu as T
    from aozora_sdxl_training_amd.unet_spec import mini_config
    model = mini_config(ctx_dim=64, pooled=32)
    cfg = T._config(tmp, "v_prediction", BATCH_SIZE=4, SAVE_EVERY_N_STEPS=2)
    if rank == 0:
        T._base_checkpoint(cfg.SINGLE_FILE_CHECKPOINT_PATH, model)
    dist.barrier()
    h, unet = _run(cfg, model)
    final = unet.pflat.clone()
    cfg2 = T._config(tmp, "v_prediction", BATCH_SIZE=4, SAVE_EVERY_N_STEPS=0, RESUME_TRAINING=True,
                     RESUME_MODEL_PATH=os.path.join(cfg.OUTPUT_DIR, "mini_run_step_2.safetensors"),
                     RESUME_STATE_PATH=os.path.join(cfg.OUTPUT_DIR, "mini_run_training_state_step_2.pt"))
    # emergency-save flag (train.py:2534-2542, 2810): one file, consumed by rank 0, acted on by EVERY rank (the save branch
    # holds collectives: a rank that skipped it would run into the next micro-step's exchange and hang)
    cfg2.FORCE_SAVE_FLAG = os.path.join(tmp, "force_save.flag")
    if rank == 0:
        open(cfg2.FORCE_SAVE_FLAG, "w").write("1")
    dist.barrier()
    h2, unet2 = _run(cfg2, model)
    out[rank] = dict(forced=h2["saved"], flag_left=os.path.exists(cfg2.FORCE_SAVE_FLAG), final=os.path.exists(h2["final_model"]),
                     losses=h["losses"], gns=h["grad_norms"], saved=h["saved"], resumed_equal=bool(torch.equal(unet2.pflat, final)),
                     resumed_losses=h2["losses"], shard=os.path.exists(cfg2.RESUME_STATE_PATH + f".rank{rank}"))
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_two_ranks_match_single_process(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import test_trainer_gpu as T
    from aozora_sdxl_training_amd.unet_spec import mini_config
    mgr = mp.Manager()
    out = mgr.dict()
    dp_dir = tmp_path / "dp"; dp_dir.mkdir()
    mp.spawn(_worker, args=(2, _port(), str(dp_dir), out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    assert r0["losses"] == r1["losses"] and r0["gns"] == r1["gns"]                # every rank reports the global values
    assert r0["saved"] and r0["shard"] and r1["shard"]
    assert r0["forced"] == r1["forced"] == [("mini_run_step_3.safetensors", "mini_run_training_state_step_3.pt")]
    assert not r0["flag_left"] and r0["final"]
    assert r0["resumed_equal"] and r1["resumed_equal"] and r0["resumed_losses"] == r0["losses"][4:]
    # single process, same config (global batch 4)
    one_dir = tmp_path / "one"; one_dir.mkdir()
    model = mini_config(ctx_dim=64, pooled=32)
    cfg = T._config(str(one_dir), "v_prediction", BATCH_SIZE=4, SAVE_EVERY_N_STEPS=0)
    T._base_checkpoint(cfg.SINGLE_FILE_CHECKPOINT_PATH, model)
    h, _ = _run(cfg, model)
    assert len(h["losses"]) == len(r0["losses"]) == 8
    for a, b in zip(r0["losses"], h["losses"]):
        assert abs(a - b) <= 2e-3 * abs(b), (r0["losses"], h["losses"])
    for a, b in zip(r0["gns"], h["grad_norms"]):
        assert abs(a - b) <= 1e-2 * b, (r0["gns"], h["grad_norms"])
