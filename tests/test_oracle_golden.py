"""Pins the ORACLE (oracle/) against vectors captured from the reference itself."""
import hashlib
import math

import torch

from oracle import step_ref as R
from oracle.unet_ref import SDXL_BASE, param_table, forward_macs

DT = {"torch.bfloat16": torch.bfloat16, "torch.float32": torch.float32}


def test_structure_matches_published_sdxl():
    t = param_table(SDXL_BASE)
    assert len(t) == 1680
    assert sum(math.prod(s) for _, s in t) == 2_567_463_684
    assert abs(forward_macs(SDXL_BASE, 128, 128) / 1e12 - 3.381) < 5e-4
    assert abs(forward_macs(SDXL_BASE, 64, 64) / 1e12 - 0.794) < 5e-4


def test_names_pass_reference_keymap(golden_host):
    # every diffusers name maps to a unique single-file SDXL key (train.py:2418-2465)
    k = golden_host["keymap"]
    assert k["n"] == 1680 and k["unique_targets"] == 1680
    assert all(v.startswith("model.diffusion_model.") for v in k["samples"].values())
    assert not any(("down_blocks" in v or "up_blocks" in v or "mid_block" in v) for v in k["samples"].values())


def test_loss_and_grad(golden_host, golden_tensors):
    for c in golden_host["loss"]:
        k = c["key"]
        curve = None if c["curve"] == "none" else golden_tensors[f"curve_{c['curve']}"]
        pred = golden_tensors[k + "_pred"].clone().requires_grad_(True)
        l = R.weighted_mse_loss(pred, golden_tensors[k + "_tgt"], golden_tensors[k + "_ts"], curve)
        l.backward()
        assert torch.equal(l.detach(), golden_tensors[k + "_loss"])
        assert torch.equal(pred.grad, golden_tensors[k + "_dpred"])


def test_survey_known_answer_loss(golden_tensors):
    torch.manual_seed(0)
    pred, tgt = torch.randn(2, 4, 8, 8), torch.randn(2, 4, 8, 8)
    ts = torch.tensor([10, 900])
    assert abs(float(R.weighted_mse_loss(pred, tgt, ts, golden_tensors["curve_flat"])) - 2.2514519691) < 1e-6
    assert abs(float(R.weighted_mse_loss(pred, tgt, ts, golden_tensors["curve_bell"])) - 0.6491934061) < 1e-6


def test_raven_math_bit_exact(golden_host, golden_tensors):
    for c in golden_host["raven"]:
        k = c["key"]
        p = golden_tensors[k + "_init"].clone()
        m = torch.zeros_like(p, dtype=DT[c["mdt"]])
        v = torch.zeros_like(p, dtype=DT[c["mdt"]])
        for s in range(c["steps"]):
            g = golden_tensors[f"{k}_g{s}"]
            R.adamw_debiased_step(p, g.float(), m, v, s + 1, c["lr"], c["betas"][0], c["betas"][1], c["eps"], c["wd"], c["debias"])
            assert torch.equal(p, golden_tensors[f"{k}_p{s}"]), (k, s)
            assert torch.equal(m, golden_tensors[f"{k}_m{s}"])
            assert torch.equal(v, golden_tensors[f"{k}_v{s}"])


def test_titan_cycle(golden_host, golden_tensors):
    for c in golden_host["titan"]:
        k = c["key"]
        assert c["grads_none_after_backward"] and c["ready_after_zero_grad"] == 0
        g1, g2 = golden_tensors[k + "_cpu_g1"].clone(), golden_tensors[k + "_cpu_g2"].clone()
        mx = float("inf") if c["max_norm"] == "inf" else c["max_norm"]
        n = R.clip_grad_norm([g1, g2], mx)
        assert torch.allclose(n, golden_tensors[k + "_norm"], rtol=1e-6)
        assert torch.allclose(g1, golden_tensors[k + "_clip_g1"], rtol=1e-6, atol=0)
        w1 = golden_tensors[k + "_w1"].clone()
        m = torch.zeros_like(w1, dtype=DT[c["mdt"]]); v = torch.zeros_like(m)
        R.adamw_debiased_step(w1, golden_tensors[k + "_clip_g1"], m, v, 1, 1e-3, 0.9, 0.999, 1e-8, 0.01, 0.3)
        assert torch.equal(w1, golden_tensors[k + "_w1_after"])
        assert torch.equal(m, golden_tensors[k + "_m1"]) and torch.equal(v, golden_tensors[k + "_v1"])


def test_clip_matches_torch(golden_tensors):
    for ci in (0, 1):
        gs = [golden_tensors[f"clip{ci}_g{i}"].clone() for i in range(3)]
        n = R.clip_grad_norm(gs, 1.0)
        assert torch.equal(n, golden_tensors[f"clip{ci}_norm"])
        for i in range(3):
            assert torch.equal(gs[i], golden_tensors[f"clip{ci}_c{i}"])


def test_error_behaviour_recorded(golden_host):
    assert golden_host["titan_double_owner"] == "RuntimeError"
    assert golden_host["raven_bad_lr"] == "ValueError" and golden_host["raven_bad_dtype"] == "ValueError"


def test_titan_two_windows_fp32_accumulation_and_clip():
    """oracle Titan restatement (titan_accumulate / titan_clip / adamw_debiased_step on fp32 gradients) against the reference's
    TitanAdamW over two accumulation windows of two micro-steps with CPU clip (tests/golden/golden_r2.*): bit-exact."""
    import json
    import os
    from oracle.step_ref import titan_accumulate, titan_clip, adamw_debiased_step
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    host = json.load(open(os.path.join(here, "golden_r2.json")))
    tens = torch.load(os.path.join(here, "golden_r2.pt"), map_location="cpu", weights_only=True)
    DT = {"torch.bfloat16": torch.bfloat16, "torch.float32": torch.float32}
    for c in host["titan_seq"]:
        k, mdt = c["key"], DT[c["mdt"]]
        mx = float("inf") if c["max_norm"] == "inf" else c["max_norm"]
        w = tens[k + "_w0"].clone()
        m, v = torch.zeros(w.shape, dtype=mdt), torch.zeros(w.shape, dtype=mdt)
        for win in range(2):
            hg = None
            for mi in range(2):
                hg = titan_accumulate(hg, tens[f"{k}_g{win}{mi}"])
            assert torch.equal(hg, tens[f"{k}_cpu{win}"])
            n = titan_clip([hg], mx)
            assert float(n) == float(tens[f"{k}_norm{win}"])
            adamw_debiased_step(w, hg, m, v, win + 1, 1e-3, 0.9, 0.999, 1e-8, 0.01, 0.3)
            assert torch.equal(w, tens[f"{k}_w{win + 1}"]) and torch.equal(m, tens[f"{k}_m{win + 1}"]) and torch.equal(v, tens[f"{k}_v{win + 1}"])
