"""Data-parallel host logic over REAL collectives (gloo, world_size 2, CPU): the flat reduce-scatter /
all-gather decomposition used by dist.ShardedRaven equals all-reduce + full update, shard bounds are
aligned, and ticket sharding reproduces the reference's global draw."""
import os
import socket

import pytest
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from aozora_sdxl_training_amd.dist import shard_bounds, intersect_ranges, reduce_scatter_flat, all_gather_flat
    from oracle.step_ref import adamw_debiased_step
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 4096
    g = torch.Generator().manual_seed(5)
    p0 = (torch.randn(n, generator=g) * 0.05).bfloat16()
    grads = [(torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) * 0.01).bfloat16() for r in range(world)]
    trainable = [(0, 1024), (1536, 4096)]                     # a frozen hole in the middle
    # --- DP path: reduce-scatter, update own trainable sub-ranges, all-gather ---
    p = p0.clone()
    gflat = grads[rank].float()           # gloo has no bf16 sum on every build: reduce in fp32, round like bf16 after
    reduce_scatter_flat(dist, gflat, rank, world)
    lo, hi = shard_bounds(n, world, rank)
    m = torch.zeros(n, dtype=torch.bfloat16); v = torch.zeros(n, dtype=torch.bfloat16)
    for a, b in intersect_ranges(trainable, lo, hi):
        adamw_debiased_step(p[a:b], gflat[a:b].bfloat16().float(), m[a:b], v[a:b], 1, 1e-3, 0.9, 0.999, 1e-8, 0.01, 0.3)
    all_gather_flat(dist, p, rank, world)
    # --- reference: all-reduce + full update on every rank ---
    q = p0.clone()
    gsum = sum(gr.float() for gr in grads).bfloat16().float()
    m2 = torch.zeros(n, dtype=torch.bfloat16); v2 = torch.zeros(n, dtype=torch.bfloat16)
    for a, b in trainable:
        adamw_debiased_step(q[a:b], gsum[a:b], m2[a:b], v2[a:b], 1, 1e-3, 0.9, 0.999, 1e-8, 0.01, 0.3)
    ok = torch.equal(p, q) and torch.equal(p[1024:1536], p0[1024:1536])
    # --- two independently sharded regions (dist.ShardedRaven's head / tail cut), collectives on region VIEWS ---
    cut = 1024 + 512                       # region lengths 1536 / 2560: both multiples of world*64
    p3 = p0.clone()
    g3 = grads[rank].float()
    m3 = torch.zeros(n, dtype=torch.bfloat16); v3 = torch.zeros(n, dtype=torch.bfloat16)
    for (ra, rb) in ((cut, n), (0, cut)):  # tail region first, as the overlapped step issues them
        reduce_scatter_flat(dist, g3[ra:rb], rank, world)
    for (ra, rb) in ((0, cut), (cut, n)):
        lo3, hi3 = shard_bounds(rb - ra, world, rank)
        for a, b in intersect_ranges(trainable, ra + lo3, ra + hi3):
            adamw_debiased_step(p3[a:b], g3[a:b].bfloat16().float(), m3[a:b], v3[a:b], 1, 1e-3, 0.9, 0.999, 1e-8, 0.01, 0.3)
    for (ra, rb) in ((0, cut), (cut, n)):
        all_gather_flat(dist, p3[ra:rb], rank, world)
    ok = ok and torch.equal(p3, q)
    out[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_update_equals_allreduce_update(world):
    """world 4: the shard arithmetic with more than one interior boundary (a rank whose shard lies wholly inside the frozen hole,
    region lengths 1536 / 2560 = multiples of world * 64); fp32 sums of four bf16 gradients are exact, so both sides round once."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert all(out[r] for r in range(world)), dict(out)


def test_shard_bounds_and_intersections():
    sys.path.insert(0, ROOT)
    from aozora_sdxl_training_amd.dist import shard_bounds, intersect_ranges
    n = 8192
    for world in (1, 2, 4, 8):
        cover = [shard_bounds(n, world, r) for r in range(world)]
        assert cover[0][0] == 0 and cover[-1][1] == n and all(cover[i][1] == cover[i + 1][0] for i in range(world - 1))
        assert all(a % 64 == 0 for a, _ in cover)
    assert intersect_ranges([(0, 100), (200, 300)], 50, 250) == [(50, 100), (200, 250)]
    try:
        shard_bounds(100, 8, 0)
        assert False
    except ValueError:
        pass


def _titan_worker(rank, world, port, out):
    """Titan under data parallel, host logic over real collectives: per-rank fp32 accumulation of the micro-step gradients,
    fp32 reduce-scatter at the window boundary, norm of the owned shard + scalar all-reduce, clip, AdamW on the owned shard with
    fp32 gradients, all-gather -- against what the REFERENCE's TitanAdamW produced in one process for the summed gradients
    (tests/golden/golden_r2.*, titan_seq: two windows x two micro-steps, CPU clip)."""
    sys.path.insert(0, ROOT)
    import json
    import torch.distributed as dist
    from aozora_sdxl_training_amd.dist import shard_bounds, reduce_scatter_flat, all_gather_flat
    from oracle.step_ref import adamw_debiased_step, titan_accumulate
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gold = os.path.join(ROOT, "tests", "golden")
    host = json.load(open(os.path.join(gold, "golden_r2.json")))
    tens = torch.load(os.path.join(gold, "golden_r2.pt"), map_location="cpu", weights_only=True)
    DT = {"torch.bfloat16": torch.bfloat16, "torch.float32": torch.float32}
    res = {}
    for c in host["titan_seq"]:
        k, mdt = c["key"], DT[c["mdt"]]
        mx = float("inf") if c["max_norm"] == "inf" else c["max_norm"]
        w = tens[k + "_w0"].clone()
        n = w.numel()
        lo, hi = shard_bounds(n, world, rank)
        m, v = torch.zeros(n, dtype=mdt), torch.zeros(n, dtype=mdt)
        worst, frac = 0.0, 0.0
        for win in range(2):
            acc = None
            for mi in range(2):      # every rank holds 1/world of the micro-step's gradient (exact in bf16: a power of two)
                acc = titan_accumulate(acc, (tens[f"{k}_g{win}{mi}"].float() / world).bfloat16())
            reduce_scatter_flat(dist, acc, rank, world)
            ss = acc[lo:hi].double().pow(2).sum().reshape(1)
            dist.all_reduce(ss)
            total = float(ss.sqrt())
            want_norm = float(tens[f"{k}_norm{win}"])
            assert abs(total - want_norm) <= 1e-6 * want_norm
            coef = min(1.0, mx / (total + 1e-6)) if mx > 0 else 1.0
            g = acc[lo:hi] * torch.tensor(coef, dtype=torch.float32) if coef < 1 else acc[lo:hi]
            adamw_debiased_step(w[lo:hi], g, m[lo:hi], v[lo:hi], win + 1, 1e-3, 0.9, 0.999, 1e-8, 0.01, 0.3)
            all_gather_flat(dist, w, rank, world)
            want = tens[f"{k}_w{win + 1}"]
            d = (w.float() - want.float()).abs()
            worst = max(worst, float((d / (want.float().abs() * 2.0 ** -7 + 1e-30)).max()))     # in bf16 ulps
            frac = max(frac, float((d > 0).float().mean()))
        res[k] = (worst, frac, mx)
    out[rank] = res
    dist.destroy_process_group()


def test_titan_sharded_fp32_accumulation_matches_reference_titan():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_titan_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        for k, (worst, frac, mx) in out[r].items():
            if mx == float("inf"):
                assert worst == 0.0 and frac == 0.0, (k, worst, frac)          # no clip: the decomposition is bit-exact
            else:
                assert worst <= 1.0 and frac <= 0.01, (k, worst, frac)        # clip: the shard-wise norm differs in the last fp32 bit
    assert out[0] == out[1]


def emulate_bf16_ring_reduce_scatter(rank_grads):
    """What RCCL's ring reduce-scatter does to bf16 gradients that the gloo stand-in (fp32 sum, ONE rounding) does not: chunk c
    travels the ring starting at rank c + 1 and every hop adds its local bf16 chunk in fp32 and sends the sum on ROUNDED to bf16,
    so the owner of chunk c holds a sum that was rounded world - 1 times.  -> the full reduced vector (all chunks, fp32 values of bf16)."""
    world = len(rank_grads)
    n = rank_grads[0].numel()
    assert n % world == 0
    out = torch.empty(n, dtype=torch.float32)
    ck = n // world
    for c in range(world):
        sl = slice(c * ck, (c + 1) * ck)
        acc = rank_grads[(c + 1) % world][sl].float()
        for h in range(2, world + 1):
            acc = (acc + rank_grads[(c + h) % world][sl].float()).bfloat16().float()
        out[sl] = acc
    return out


def test_bf16_ring_rounding_against_the_one_rounding_sum():
    """ShardedRaven's 2-rank "equals the single process" tests run over gloo (fp32 sum, one rounding).  Under
    RCCL at N = 8 the bf16 ring rounds at every hop.  Emulated here on the per-rank gradients of a mini SDXL-topology UNet (oracle,
    bf16 autocast, one micro-batch per rank, loss pre-scaled by 1 / world as TrainStep does).  Measured (1.5 M elements): the GLOBAL
    NORM -- the observable north_star bounds at 1e-3 -- deviates from the exact sum's by 1.7e-4 (1.3e-5 with one rounding), single
    elements carry twice the rounding noise of the one-rounding sum (relative L2 of the difference 3.4e-3 vs 1.7e-3, the size of
    the bf16 storage noise the gradients carry anyway).  A sixth of the 1e-3 budget, not a threat to it: the bf16 exchange stays
    (half the wire bytes of an fp32 one); ShardedTitan's exchange is fp32 and insensitive to this."""
    sys.path.insert(0, ROOT)
    import math
    from oracle.unet_ref import UNetConfig as OC, init_params
    from oracle.step_ref import RefTrainer
    world = 8
    oc = OC(block_out_channels=(32, 64), transformer_layers=(0, 1), head_dim=32, cross_attention_dim=64, addition_time_embed_dim=32,
            pooled_dim=32, norm_groups=8)
    params = {k: v.bfloat16().float() for k, v in init_params(oc, seed=7).items()}
    g = torch.Generator().manual_seed(3)
    grads = []
    for r in range(world):
        tr = RefTrainer(oc, params, mode="epsilon", bf16=True, ga=world, clip=1.0)        # ga = world: the 1 / world pre-scaling of the loss seed
        lat = torch.randn(2, 4, 8, 8, generator=g).bfloat16(); noise = torch.randn(2, 4, 8, 8, generator=g)
        ctx = torch.randn(2, 7, 64, generator=g).bfloat16(); pooled = torch.randn(2, 32, generator=g).bfloat16()
        tid = torch.tensor([[64, 64, 0, 0, 64, 64]] * 2, dtype=torch.bfloat16)
        tr.micro_step(lat, noise, torch.tensor([100 + 97 * r, 900 - 83 * r]), ctx, pooled, tid)
        flat = torch.cat([v.detach().reshape(-1) for _, v in sorted(tr.grads().items())]).bfloat16()
        pad = (-flat.numel()) % world
        grads.append(torch.cat([flat, torch.zeros(pad, dtype=torch.bfloat16)]))
    exact = sum(gr.double() for gr in grads)
    one = sum(gr.float() for gr in grads).bfloat16().float()                 # gloo stand-in / single process: fp32 sum, one rounding
    ring = emulate_bf16_ring_reduce_scatter(grads)
    nrm = lambda t: float(t.double().norm())
    dev_norm_ring = abs(nrm(ring) - nrm(exact)) / nrm(exact)
    dev_norm_one = abs(nrm(one) - nrm(exact)) / nrm(exact)
    rel_ring = nrm(ring.double() - exact) / nrm(exact)
    rel_one = nrm(one.double() - exact) / nrm(exact)
    print(f"bf16 ring emulation, {world} ranks, {grads[0].numel()} elements: global-norm deviation ring {dev_norm_ring:.2e} / one rounding {dev_norm_one:.2e}; "
          f"element-wise relative L2 vs the exact sum: ring {rel_ring:.2e} / one rounding {rel_one:.2e}")
    assert dev_norm_ring <= 3e-4                     # measured 1.7e-4: a sixth of the 1e-3 bar on the gradient norm
    assert rel_ring <= 3.0 * rel_one + 1e-6 and rel_ring <= 6e-3
    assert math.isfinite(rel_one) and rel_one > 0
