"""`python -m aozora_sdxl_training_amd.trainer --config X.json` -- the process the unmodified GUI would spawn in place of
train.py (gui/gui.py:5930-5975): a nested GUI preset on disk -> TrainingConfig -> cache -> HIP steps -> the reporter's stdout
lines, parsed here with the three regular expressions of the GUI's live-metrics widget (gui/gui.py:1856-1878) -> final model."""
import dataclasses
import json
import os
import re
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# the GUI's own patterns (data of the wire protocol between trainer and GUI)
RE_PROGRESS = re.compile(r'Training\s*\|.*\|\s*(\d+)/(\d+)\s*\[.*?\]\s*\[Loss:\s*([\d.e+-]+),\s*Ticket:\s*(\d+),\s*Sigma:\s*([\d.e+-]+)\]')
RE_OPTIM = re.compile(r'--- Optimizer Step:\s*(\d+)\s*\|\s*Loss:\s*([\d.e+-]+)\s*\|\s*LR:\s*([\d.e+-]+)\s*---')
RE_GRAD = re.compile(r'Grad Norm \(Raw/Clipped\):\s*([\d.]+)\s*/\s*([\d.]+)')


def test_entry_point_drives_a_run_from_a_gui_preset(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import test_trainer_gpu as T
    from aozora_sdxl_training_amd.unet_spec import mini_config
    from aozora_sdxl_training_amd import config as C
    model = mini_config(ctx_dim=64, pooled=32)
    tmp = str(tmp_path)
    cfg = T._config(tmp, "v_prediction", SAVE_EVERY_N_STEPS=0, OUTPUT_NAME="entry_{uuid}")
    T._base_checkpoint(cfg.SINGLE_FILE_CHECKPOINT_PATH, model)
    # the flat test config as the nested preset the GUI writes (strings where the GUI's widgets hand over strings)
    nested = {nk: getattr(cfg, fk) for nk, fk in C.block_keys("sdxl").items() if hasattr(cfg, fk)}
    nested["sdxl_max_train_steps"] = "8"
    nested["sdxl_raven_params"] = dict(cfg.RAVEN_PARAMS, betas=list(cfg.RAVEN_PARAMS["betas"]))
    preset = tmp_path / "preset.json"
    preset.write_text(json.dumps({"config_version": 5, "active_mode": "sdxl", "sdxl": nested}))
    spec = tmp_path / "unet.json"
    spec.write_text(json.dumps(dataclasses.asdict(model)))
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-u", "-m", "aozora_sdxl_training_amd.trainer", "--config", str(preset), "--unet-config", str(spec)],
                       cwd=tmp, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = r.stdout
    assert f"INFO: Loading configuration from {preset}" in out and "INFO: Set random seed to 42" in out
    assert "--- STARTING STANDARD SDXL TRAINING ---" in out and "Training complete." in out and "All tasks complete. Final model saved." in out
    prog = [m for m in (RE_PROGRESS.search(line) for line in re.split(r"[\r\n]", out)) if m]
    # the reference prints micro_step + 1 for its 1-based micro_step (train.py:2713, 439); the GUI subtracts one (gui.py:1861)
    assert [int(m.group(1)) for m in prog][-1] == 9 and all(int(m.group(2)) == 8 for m in prog)
    assert all(0.0 < float(m.group(3)) < 10.0 and 0 <= int(m.group(4)) < 1000 and 0.0 <= float(m.group(5)) <= 1.0 for m in prog)
    opt = RE_OPTIM.findall(out)
    assert [int(o[0]) for o in opt] == [1, 2, 3, 4] and float(opt[-1][2]) == pytest.approx(2e-5, rel=1e-3)
    grads = RE_GRAD.findall(out)
    assert len(grads) == 4 and all(float(a) > 0 and float(b) <= 1.0 + 1e-6 for a, b in grads)
    finals = [f for f in os.listdir(cfg.OUTPUT_DIR) if re.fullmatch(r"entry_[a-z0-9]{6}\.safetensors", f)]
    assert len(finals) == 1
    # an Anima preset is refused, not attempted
    preset2 = tmp_path / "anima.json"
    preset2.write_text(json.dumps({"active_mode": "anima"}))
    r2 = subprocess.run([sys.executable, "-m", "aozora_sdxl_training_amd.trainer", "--config", str(preset2)], cwd=tmp, env=env,
                        capture_output=True, text=True, timeout=300)
    assert r2.returncode == 2 and "Anima DiT" in r2.stdout


def test_bench_started_bare_with_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` with no launcher around it (the form the driver uses for N = 1) becomes the launcher itself:
    two fresh rank processes, rendezvous on 127.0.0.1, ONE JSON line with n_gpus = 2 relayed from rank 0.  Rehearsed here on a
    one-GPU box: --rehearse-gloo puts both ranks on cuda:0 with the gloo backend and the mini UNet (control flow only)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-gloo", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-live-pmc", "--through-trainer", "0"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["metric"].startswith("SDXL UNet train iters/sec")
    assert out["config"]["parallelism"] == "dp2" and out["value"] > 0 and len(out["exchange"]["per_rank"]) == 2
    assert isinstance(out["hbm_roofline"], list) and all(0 < x["frac"] < 1.5 for x in out["hbm_roofline"])
    # every rank reports its exchange anatomy AND where it sits on the host: its own share of the CPUs (affinity.bind_rank ran before
    # the pinned m / v shards were allocated), intra-op threads capped
    per_rank = out["exchange"]["per_rank"]
    assert all(isinstance(pr, dict) and "optimizer_boundary_on_main_stream" in pr and "mv_h2d" not in pr for pr in per_rank), per_rank
    place = [pr["host_placement"] for pr in per_rank]
    assert [pl["local_rank"] for pl in place] == [0, 1] and all(pl["local_world"] == 2 and 1 <= pl["threads"] <= 8 for pl in place), place
    if len(os.sched_getaffinity(0)) >= 2:
        assert place[0]["cpus"] != place[1]["cpus"] and all(pl["n_cpus"] <= len(os.sched_getaffinity(0)) // 2 for pl in place), place
    # the m / v copy streams made their first copies before the process group existed (streams.host_link_streams; with the nccl backend
    # a copy stream first used after the communicator loses its SDMA engine: profiles/r04_host_link_and_rccl.txt)
    assert "host-link streams: first copies made before any RCCL communicator exists" in r.stderr, r.stderr[-2000:]
    # a launcher that sets a different world size is refused, not silently mis-counted
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-gloo"], cwd=ROOT,
                        env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert r2.returncode != 0 and "WORLD_SIZE=1" in (r2.stderr + r2.stdout)
