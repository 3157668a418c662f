"""SURVEY 8f rows f1 + f2 (data feed, cache reader): this repo's aozora_sdxl_training_amd/data.py replayed over synthetic
caches (tests/golden/synth_cache.py) must reproduce what the REFERENCE's dataset / samplers / schedules / collate
produced over identically built caches (tests/golden/golden_data.json, written by tests/golden/make_golden_data.py, which
imports the reference).  Integer results (orders, batches, packed ids, chosen files, shapes) bit-exact; tensor sums exact
(the feed only moves and linearly mixes bf16 tensors; sums are taken in float64 on both sides)."""
import json
import os
import random
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import synth_cache                                             # noqa: E402
from make_golden_data import CONFIGS, make_config              # noqa: E402  (pure helpers; the reference is NOT imported here)
from aozora_sdxl_training_amd import data as D                 # noqa: E402
from aozora_sdxl_training_amd.schedule import build_timestep_ticket_pool   # noqa: E402

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_data.json")))


def _describe(item):
    if item is None:
        return None
    f = lambda t: float(t.double().sum())
    return dict(image_key=item["image_key"].replace(os.sep, "/"), te=os.path.basename(item["latent_path"]),
                embeds_shape=list(item["embeds"].shape), embeds_dtype=str(item["embeds"].dtype), embeds_sum=f(item["embeds"]),
                pooled_shape=list(item["pooled"].shape), pooled_sum=f(item["pooled"]), latents_shape=list(item["latents"].shape),
                latents_sum=f(item["latents"]), original=list(item["original_sizes"]), scaled=list(item["scaled_sizes"]),
                target=list(item["target_sizes"]), crop=list(item["crop_coords"]))


@pytest.mark.parametrize("name", list(CONFIGS))
def test_feed_matches_reference(name, tmp_path, capsys):
    spec, gold = CONFIGS[name], GOLD[name]
    cfg = make_config(spec, str(tmp_path))
    ds = D.CachedLatentDataset(cfg)
    assert len(ds) == gold["n"]
    assert [[it["relative_path"].replace(os.sep, "/"), list(it["target_size"])] for it in ds.items] == gold["order"]
    assert ds.dropout_prob == gold["dropout_prob"] and [ds.cond_scale_min, ds.cond_scale_max] == gold["cond"]
    # bucket sampler: two consecutive epochs, resume offset, length
    for key, g in gold["bucket_sampler"].items():
        bs, mode = int(key[2:].split("_")[0]), key.split("_")[1]
        s = D.BucketBatchSampler(ds, bs, spec["SEED"], shuffle=(mode == "shuf"))
        assert [list(b) for b in s] == g["epoch0"] and [list(b) for b in s] == g["epoch1"] and len(s) == g["length"]
        s2 = D.BucketBatchSampler(ds, bs, spec["SEED"], shuffle=(mode == "shuf"))
        s2.set_epoch(1); s2.set_start_batch_index(2)
        assert [list(b) for b in s2] == g["epoch1_from2"]
        for b in g["epoch0"]:
            assert len({ds.bucket_keys[i] for i in b}) == 1 or bs == 1          # one resolution per batch
    # schedules
    steps, bs = 14, 3
    pool, bin_ranges = build_timestep_ticket_pool({"bin_size": 100, "counts": [45, 143, 176, 173, 154, 126, 94, 59, 26, 4]}, steps * bs, 1000, spec["SEED"], False)
    gs = gold["schedule"]
    for spread in (False, True):
        raw = D.batch_schedule(ds, steps, bs, spec["SEED"], pool, bin_ranges, spread)
        k = "spread" if spread else "epoch"
        assert [list(map(int, b)) for b in raw] == gs[k]["raw"]
        assert D.pack_schedule(raw, bs) == gs[k]["packed"]
    assert [list(map(int, b)) for b in D.batch_schedule(ds, 30, 1, spec["SEED"], pool[:30], bin_ranges, True)] == gs["spread_bs1"]
    assert D.image_schedule(len(ds), 40, spec["SEED"], pool[:40], bin_ranges, False).tolist() == gs["image_epoch"]
    assert D.timestep_bin_ids(pool[:20], bin_ranges).tolist() == gs["bin_ids"]
    ps = D.PrecomputedBatchSampler(gs["epoch"]["packed"], spec["SEED"], 3)
    assert dict(batches=[b for b in ps], length=len(ps), epoch=ps.epoch) == gold["precomputed_from3"]
    # samples and collated batches through the same ids the reference served
    n_none = 0
    for g in gold["batches"]:
        got = [ds[i] for i in g["ids"]]
        assert [_describe(x) for x in got] == g["items"]
        n_none += sum(x is None for x in got)
        if g["collate"] is not None:
            col = D.collate(got)
            seen = {k: (list(v.shape) if torch.is_tensor(v) else (v if k != "latent_path" else [os.path.basename(p) for p in v]))
                    for k, v in col.items() if k != "image_key"}
            seen = json.loads(json.dumps(seen))                     # tuples -> lists, like the golden
            assert seen == g["collate"]
    assert D.pack_sample_index(3, 0) == gold["pack"][0] and D.pack_sample_index(7, 123456) == gold["pack"][1]
    assert list(D.unpack_sample_index((99 << 32) | 12)) == gold["pack"][2]
    capsys.readouterr()


def test_cache_helpers_match_reference():
    g = GOLD["cache_helpers"]
    for seed, expect in g["caption_choice"]:
        w = {"tags": 40, "nl": 10, "tags_nl": 25, "nl_tags": 25} if seed % 2 == 0 else {"tags": 0, "nl": 0, "tags_nl": 0, "nl_tags": 5}
        assert D.choose_caption_variant(random.Random(seed), w) == expect
    assert [D.item_stem_from_te_path("/x/a_b_mb2_json_tags_nl_te.pt"), D.base_stem_from_cache_path("/x/a_b_mb2_json_tags_nl_te.pt"),
            D.base_stem_from_cache_path("/x/a_b_mb3_lat.pt"), str(D.lat_path_for_te_path("/x/a_b_json_nl_te.pt")).replace(os.sep, "/")] == g["stems"][:4]
    assert [D.caption_source_type("JSON "), D.caption_source_type(None), D.caption_source_type("weird")] == g["caption_source"]
    assert [D.json_caption_weights(types.SimpleNamespace()), D.json_caption_weights(types.SimpleNamespace(
        CAPTION_TAGS_PERCENT=0, CAPTION_NL_PERCENT=0, CAPTION_TAGS_NL_PERCENT=0, CAPTION_NL_TAGS_PERCENT=-3))] == g["weights"]
    assert [list(D.text_conditioning_scale_range(types.SimpleNamespace(TEXT_CONDITIONING_SCALE_ENABLED=True, TEXT_CONDITIONING_SCALE_MIN=1.7, TEXT_CONDITIONING_SCALE_MAX=0.2))),
            list(D.text_conditioning_scale_range(types.SimpleNamespace()))] == g["scale_range"]
    assert D.cache_folder_name(True).endswith("_rf") and D.cache_folder_name(False).endswith("_standard_sdxl")


def test_dataloader_and_rank_sharding(tmp_path, capsys):
    """The consumer of train.py:2655-2658 (DataLoader + batch sampler + collate) and the data-parallel row slicing."""
    cfg = make_config(CONFIGS["two_sets_repeats"], str(tmp_path))
    ds = D.CachedLatentDataset(cfg)
    sched = D.pack_schedule(D.batch_schedule(ds, 6, 4, 7, [0] * 24, [(0, 1000)], False), 4)
    dl = torch.utils.data.DataLoader(ds, batch_sampler=D.PrecomputedBatchSampler(sched, 7, 1), collate_fn=D.collate, num_workers=0)
    batches = list(dl)
    assert len(batches) == 5
    full = [b for b in batches if b and b["latents"].shape[0] == 4][0]
    parts = [D.shard_batch(full, r, 2) for r in range(2)]
    assert torch.equal(torch.cat([p["latents"] for p in parts]), full["latents"])
    assert parts[0]["target_sizes"] + parts[1]["target_sizes"] == full["target_sizes"]
    with pytest.raises(ValueError):
        D.shard_batch(full, 0, 3)
    capsys.readouterr()
