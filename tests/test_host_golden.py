"""Host-side schedule logic (product code) vs vectors captured from the reference
(tests/golden/make_golden.py). Integer work: bit-exact. Python floats: exact."""
import types

import torch

from aozora_sdxl_training_amd import schedule as S


def _chk(pool):
    return int(sum((i + 1) * t for i, t in enumerate(pool)) % (2 ** 61 - 1))


def test_ticket_pools_bit_exact(golden_host):
    for c in golden_host["tickets"]:
        pool, ranges = S.build_timestep_ticket_pool(c["allocation"], c["total"], 1000, c["seed"], c["stratified"])
        assert len(pool) == c["length"]
        assert pool[:64] == c["head"] and pool[-8:] == c["tail"]
        assert _chk(pool) == c["checksum"]
        assert [list(r) for r in ranges] == c["ranges"]
        assert [sum(1 for t in pool if lo <= t < hi) for lo, hi in ranges] == c["hist"]


def test_ticket_known_answer_from_survey():
    pool, _ = S.build_timestep_ticket_pool(None, 16, 1000, 42, False)
    assert pool == [369, 409, 285, 308, 871, 552, 673, 420, 77, 165, 776, 978, 243, 597, 8, 143]


def test_sampler_sequence_and_resume(golden_host):
    g = golden_host["sampler"]
    cfg = types.SimpleNamespace(MAX_TRAIN_STEPS=10, BATCH_SIZE=4, SEED=42, is_rectified_flow=False,
                                TIMESTEP_ALLOCATION={"bin_size": 100, "counts": [45, 143, 176, 173, 154, 126, 94, 59, 26, 4]},
                                TIMESTEP_STRATIFIED_SAMPLING=False)
    s = S.TimestepSampler(cfg, "cpu")
    seq = [s.sample(4)[0].tolist() for _ in range(12)]
    assert seq == g["seq"]
    s.set_current_step(3)
    assert s.sample(4)[0].tolist() == g["after_set3"]
    assert s.state_dict() == g["state"]


def test_sampler_dp_shards_equal_global_draw():
    cfg = types.SimpleNamespace(MAX_TRAIN_STEPS=6, BATCH_SIZE=8, SEED=42, TIMESTEP_ALLOCATION=None)
    ref = S.TimestepSampler(cfg)
    shards = [S.TimestepSampler(cfg) for _ in range(4)]
    for _ in range(6):
        g = ref.sample(8)[0].tolist()
        got = sum((sh.sample_shard(8, r, 4)[0].tolist() for r, sh in enumerate(shards)), [])
        assert got == g


def test_lr_curve_exact(golden_host):
    class Opt:
        def __init__(self):
            self.param_groups = [{"lr": 0.0, "lr_scale": 1.0}, {"lr": 0.0, "lr_scale": 0.5}]
    for c in golden_host["lr"]:
        o = Opt()
        sch = S.CustomCurveLRScheduler(o, [list(p) for p in c["curve"]], c["total"])
        for ms, want in zip(range(0, c["total"] + 2), c["lrs"]):
            sch.step(ms)
            assert [g["lr"] for g in o.param_groups] == want
    o = Opt()
    sch = S.CustomCurveLRScheduler(o, [[0.0, 0.0], [0.05, 8.0e-7], [0.85, 8.0e-7], [1.0, 1.0e-7]], 100)
    sch.step(3)
    assert o.param_groups[0]["lr"] == 4.848484848484849e-07


def test_noise_and_jitter_streams(golden_host, golden_tensors):
    g = torch.Generator(device="cpu")
    for c in golden_host["rng"]["noise"]:
        n = S.generate_noise(torch.zeros(c["shape"]), g, "cpu", step=c["step"], seed=c["seed"])
        assert torch.equal(n, golden_tensors[c["key"]])
    for c in golden_host["rng"]["jitter"]:
        gen = S.seeded_torch_generator("cpu", c["seed"], *c["parts"])
        assert int(gen.initial_seed()) == c["initial_seed"]
        assert torch.equal(torch.rand((c["n"],), dtype=torch.float32, generator=gen), golden_tensors[c["key"]])


def test_loss_weight_curves(golden_tensors):
    mk = lambda v: types.SimpleNamespace(TIMESTEP_LOSS_WEIGHT_CURVE=v)
    assert torch.equal(S.timestep_loss_curve_from_config(mk([[0.0, 1.0], [1.0, 1.0]]), 1000), golden_tensors["curve_flat"])
    assert torch.equal(S.timestep_loss_curve_from_config(mk({"preset": "bell"}), 1000), golden_tensors["curve_bell"])
    assert torch.equal(S.timestep_loss_curve_from_config(mk([[0.1, 0.5], [0.5, 2.0], [0.9, 0.25]]), 1000), golden_tensors["curve_custom"])
    assert torch.equal(S.timestep_loss_curve_from_config(mk(None), 1000), golden_tensors["curve_none"])


def test_time_ids_bf16_rounding():
    t = S.make_time_ids([(805, 1024)], [(0, 0)], [(805, 1024)])
    assert t.dtype == torch.bfloat16 and t.tolist() == [[1024.0, 804.0, 0.0, 0.0, 1024.0, 804.0]]


def test_freeze_masks(golden_host):
    from aozora_sdxl_training_amd.unet_spec import param_table, SDXL_BASE
    import math
    table = param_table(SDXL_BASE)
    names = [n for n, _ in table]
    for c in golden_host["freeze"]:
        mask = S.trainable_mask(names, c["keywords"])
        frozen = [n for n, m in zip(names, mask) if not m]
        assert len(frozen) == c["n_frozen"]
        assert sum(math.prod(s) for (n, s), m in zip(table, mask) if not m) == c["frozen_numel"]
        assert frozen[:3] == c["first"] and frozen[-3:] == c["last"]
