"""Golden vectors for the data feed: run the REFERENCE's dataset, samplers, schedules and
collate (imported exactly like make_golden.py does) over synthetic caches built by synth_cache.py and record what they
produce.  Run in the build container only:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden_data.py

Writes tests/golden/golden_data.json.  Tensors are recorded as exact float64 sums / shapes / dtypes (the feed only
moves and linearly mixes cached tensors), paths as basenames (the cache lives in a temp directory)."""
import json
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import synth_cache                                  # noqa: E402
from make_golden import import_reference            # noqa: E402

CONFIGS = {
    "plain": dict(SEED=42, is_rectified_flow=False, json=False, datasets=[dict(n=23, seed=0, repeats=1)]),
    "two_sets_repeats": dict(SEED=7, is_rectified_flow=True, json=False, datasets=[dict(n=9, seed=1, repeats=2, tag="a"), dict(n=6, seed=2, repeats=1, tag="b")]),
    "json_dropout_scale": dict(SEED=1234, is_rectified_flow=False, json=True, datasets=[dict(n=17, seed=3, repeats=1, chunked_every=4)],
                               UNCONDITIONAL_DROPOUT=True, UNCONDITIONAL_DROPOUT_CHANCE=0.3,
                               TEXT_CONDITIONING_SCALE_ENABLED=True, TEXT_CONDITIONING_SCALE_MIN=0.6, TEXT_CONDITIONING_SCALE_MAX=1.4,
                               CAPTION_TAGS_PERCENT=40, CAPTION_NL_PERCENT=10, CAPTION_TAGS_NL_PERCENT=25, CAPTION_NL_TAGS_PERCENT=25),
    "dropout_only": dict(SEED=5, is_rectified_flow=False, json=False, datasets=[dict(n=12, seed=4, repeats=1, chunked_every=3)],
                         UNCONDITIONAL_DROPOUT=True, UNCONDITIONAL_DROPOUT_CHANCE=0.5),
}


TELEMETRY_CASES = [
    dict(global_step=0, timing=dict(raw_step_time=1.2345, elapsed_time=3.0, eta=4000.7, loss=0.123456, timestep="512", sigma=0.51234567), diag=None),
    dict(global_step=499, timing=dict(raw_step_time=0.1749, elapsed_time=87.4, eta=87.6, loss=1.0, timestep="7"), diag=None),
    dict(global_step=31, timing=dict(raw_step_time=2.0, elapsed_time=64.0, eta=float("nan"), loss=0.09876, timestep="999", sigma=1.0),
         diag=dict(optim_step=4, avg_loss=0.1234567, current_lr=8e-7, raw_grad_norm=3.50031, clipped_grad_norm=1.0, update_delta=1.0,
                   optim_step_time=1.18, avg_optim_step_time=1.2049)),
    dict(global_step=999, timing=dict(), diag=dict(optim_step=125, avg_loss=0.05, current_lr=1e-7, raw_grad_norm=0.0, clipped_grad_norm=0.0,
                                                  update_delta=0.0, optim_step_time=12.5, avg_optim_step_time=13.0)),
]


def make_config(spec, tmp):
    ds = []
    for i, d in enumerate(spec["datasets"]):
        root = os.path.join(tmp, f"set{i}")
        synth_cache.build(root, n_items=d["n"], json_mode=spec["json"], seed=d["seed"], rf=spec["is_rectified_flow"],
                          chunked_every=d.get("chunked_every", 0), tag=d.get("tag", ""))
        ds.append({"path": root, "repeats": d["repeats"]})
    cfg = types.SimpleNamespace(INSTANCE_DATASETS=ds, CAPTION_SOURCE_TYPE="json" if spec["json"] else "txt")
    for k, v in spec.items():
        if k not in ("json", "datasets"):
            setattr(cfg, k, v)
    return cfg


def tsum(t):
    import torch
    return None if t is None else float(t.double().sum())


def describe_item(item):
    if item is None:
        return None
    return dict(image_key=item["image_key"].replace(os.sep, "/"), te=os.path.basename(item["latent_path"]),
                embeds_shape=list(item["embeds"].shape), embeds_dtype=str(item["embeds"].dtype), embeds_sum=tsum(item["embeds"]),
                pooled_shape=list(item["pooled"].shape), pooled_sum=tsum(item["pooled"]), latents_shape=list(item["latents"].shape),
                latents_sum=tsum(item["latents"]), original=list(item["original_sizes"]), scaled=list(item["scaled_sizes"]),
                target=list(item["target_sizes"]), crop=list(item["crop_coords"]))


def main():
    import numpy as np
    import torch
    train, _, _ = import_reference()
    out = {}
    for name, spec in CONFIGS.items():
        with tempfile.TemporaryDirectory() as tmp:
            cfg = make_config(spec, tmp)
            ds = train.ImageTextLatentDataset(cfg)
            rec = dict(n=len(ds), order=[[it["relative_path"].replace(os.sep, "/"), list(it["target_size"])] for it in ds.items],
                       dropout_prob=ds.dropout_prob, cond=[ds.cond_scale_min, ds.cond_scale_max])
            # samplers
            samp = {}
            for bs, shuffle in ((1, True), (3, True), (3, False), (4, True)):
                s = train.BucketBatchSampler(ds, bs, spec["SEED"], shuffle=shuffle)
                ep0 = [list(map(int, b)) for b in s]
                ep1 = [list(map(int, b)) for b in s]
                s2 = train.BucketBatchSampler(ds, bs, spec["SEED"], shuffle=shuffle)
                s2.set_epoch(1); s2.set_start_batch_index(2)
                samp[f"bs{bs}_{'shuf' if shuffle else 'seq'}"] = dict(epoch0=ep0, epoch1=ep1, epoch1_from2=[list(map(int, b)) for b in s2], length=len(s))
            rec["bucket_sampler"] = samp
            # schedules (train.py:688-882) with a real ticket pool
            steps, bs = 14, 3
            pool, bin_ranges = train.build_timestep_ticket_pool({"bin_size": 100, "counts": [45, 143, 176, 173, 154, 126, 94, 59, 26, 4]},
                                                                steps * bs, 1000, spec["SEED"], False)
            sched = {}
            for spread in (False, True):
                raw = train.build_image_batch_schedule(ds, steps, bs, spec["SEED"], pool, bin_ranges, spread)
                packed = train.pack_sdxl_sample_schedule(raw, bs)
                sched["spread" if spread else "epoch"] = dict(raw=[list(map(int, b)) for b in raw], packed=[[int(x) for x in b] for b in packed])
            raw1 = train.build_image_batch_schedule(ds, 30, 1, spec["SEED"], pool[:30], bin_ranges, True)
            sched["spread_bs1"] = [list(map(int, b)) for b in raw1]
            sched["image_epoch"] = train.build_image_schedule(len(ds), 40, spec["SEED"], pool[:40], bin_ranges, False).tolist()
            sched["bin_ids"] = train.timestep_bin_ids(pool[:20], bin_ranges).tolist()
            rec["schedule"] = sched
            ps = train.PrecomputedImageBatchSampler(sched["epoch"]["packed"], spec["SEED"], 3)
            rec["precomputed_from3"] = dict(batches=[b for b in ps], length=len(ps), epoch=ps.epoch)
            # items + collate through the reference's own batch path
            items = []
            for batch in sched["spread"]["packed"][:8]:
                got = [ds[i] for i in batch]
                col = train.custom_collate_fn(got) if all(g is None or g["embeds"].shape == next(x for x in got if x is not None)["embeds"].shape for g in got) else None
                items.append(dict(ids=batch, items=[describe_item(g) for g in got],
                                  collate=None if not col else {k: (list(v.shape) if torch.is_tensor(v) else (v if k != "latent_path" else [os.path.basename(p) for p in v]))
                                                                for k, v in col.items() if k not in ("image_key",)}))
            rec["batches"] = items
            rec["pack"] = [train.ImageTextLatentDataset.pack_sample_index(3, 0), train.ImageTextLatentDataset.pack_sample_index(7, 123456),
                           list(train.ImageTextLatentDataset.unpack_sample_index((99 << 32) | 12))]
            out[name] = rec
    # cache.py helpers (f2)
    import training_utils.caching.cache as rc
    rng_cases = []
    import random
    for seed in (0, 1, 2, 3, 4, 5, 6, 7):
        rng = random.Random(seed)
        w = {"tags": 40, "nl": 10, "tags_nl": 25, "nl_tags": 25} if seed % 2 == 0 else {"tags": 0, "nl": 0, "tags_nl": 0, "nl_tags": 5}
        rng_cases.append([seed, rc.choose_caption_variant(rng, w)])
    out["cache_helpers"] = dict(
        caption_choice=rng_cases,
        stems=[rc.cache_item_stem_from_te_path("/x/a_b_mb2_json_tags_nl_te.pt"), rc.cache_base_stem_from_te_path("/x/a_b_mb2_json_tags_nl_te.pt"),
               rc.cache_base_stem_from_cache_path("/x/a_b_mb3_lat.pt"), str(rc.lat_path_for_te_path("/x/a_b_json_nl_te.pt")).replace(os.sep, "/"),
               rc.cache_stem_for_image("/d", "/d/sub/im.png")],
        caption_source=[rc.caption_source_type("JSON "), rc.caption_source_type(None), rc.caption_source_type("weird")],
        weights=[train.get_json_caption_weights(types.SimpleNamespace()), train.get_json_caption_weights(types.SimpleNamespace(
            CAPTION_TAGS_PERCENT=0, CAPTION_NL_PERCENT=0, CAPTION_TAGS_NL_PERCENT=0, CAPTION_NL_TAGS_PERCENT=-3))],
        scale_range=[list(train.get_text_conditioning_scale_range(types.SimpleNamespace(TEXT_CONDITIONING_SCALE_ENABLED=True, TEXT_CONDITIONING_SCALE_MIN=1.7, TEXT_CONDITIONING_SCALE_MAX=0.2))),
                     list(train.get_text_conditioning_scale_range(types.SimpleNamespace()))],
    )
    # telemetry line formats (f4): the reference reporter's own handlers, stdout captured
    import contextlib
    import io
    rep = train.AsyncReporter(total_steps=1000, test_param_name="conv_in")
    tele = []
    for case in TELEMETRY_CASES:
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            rep._last_line_len = 0
            rep._handle_log_step(case["global_step"], case["timing"], case["diag"])
        tele.append(buf.getvalue())
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rep._last_line_len = 17
        rep._handle_message("hello")
    tele.append(buf.getvalue())
    out["telemetry"] = dict(lines=tele, times=[rep._format_time(x) for x in (None, float("inf"), 0, 59.9, 3600, 86399, 360000)])
    rep.stop_event.set()
    with open(os.path.join(HERE, "golden_data.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote golden_data.json:", {k: (v["n"] if "n" in v else "-") for k, v in out.items()})


if __name__ == "__main__":
    main()
