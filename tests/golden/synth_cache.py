"""Deterministic synthetic SDXL embedding/latent cache in the reference's on-disk format (cache schema v13:
`<dataset>/.precomputed_embeddings_cache_standard_sdxl/{*_te.pt, *_lat.pt, dataset_index.pt, null_embeds.pt}`;
payload / index fields as written by train.py:1803-1830, 1966-1986).  Shared by tests/golden/make_golden.py (which
runs the REFERENCE dataset / samplers over it) and tests/test_data_feed.py (which runs this repo's data feed over an
identically built copy): the cache itself never needs to be committed.

Small tensor sizes on purpose (embeds 77 x 64, pooled 32, latents 4 x h/8 x w/8 of tiny buckets): the data feed is
shape-agnostic."""
import os
from pathlib import Path

import torch

BUCKETS = [(128, 128), (96, 160), (160, 96)]           # (w, h)
CAPTION_TYPES = ("tags", "nl", "tags_nl", "nl_tags")
CACHE_DIR = ".precomputed_embeddings_cache_standard_sdxl"
CACHE_DIR_RF = ".precomputed_embeddings_cache_rf"


def build(root, n_items=23, json_mode=False, seed=0, rf=False, chunked_every=0, tag=""):
    """-> cache dir.  Item k: relative_path sub{k%3}/{tag}Img_{k:03d}.png, bucket BUCKETS[(k*k+seed) % 3]; every
    `chunked_every`-th item carries 154-token embeddings (caption chunking)."""
    root = Path(root)
    cache = root / (CACHE_DIR_RF if rf else CACHE_DIR)
    cache.mkdir(parents=True, exist_ok=True)
    g = torch.Generator().manual_seed(1000 + seed)
    files = []
    for k in range(n_items):
        w, h = BUCKETS[(k * k + seed) % 3]
        rel = os.path.join(f"sub{k % 3}", f"{tag}Img_{k:03d}.png")
        stem = rel[:-4].replace(os.sep, "_")
        ntok = 154 if (chunked_every and k % chunked_every == 0) else 77
        meta = dict(relative_path=rel, original_size=(w * 2 + k, h * 2 + 3), scaled_size=(w + (k % 5), h + (k % 3)), target_size=(w, h),
                    crop_coords=(k % 4, (k * 3) % 7), bucket_variant_index=0)
        lat = cache / f"{stem}_lat.pt"
        latents = torch.randn(4, h // 8, w // 8, generator=g).to(torch.bfloat16)
        if k == 5:
            latents[0, 0, 0] = float("nan")           # the reference drops such samples (train.py:2133)
        torch.save({"latents": latents, "cache_options": {"cache_schema_version": 13}}, lat)
        variants = {}
        for ct in (CAPTION_TYPES if json_mode else ("txt",)):
            suffix = f"_json_{ct}" if json_mode else ""
            te = cache / f"{stem}{suffix}_te.pt"
            payload = dict(meta, original_stem=Path(rel).stem, caption_type=ct, caption=f"caption {k} {ct}",
                           embeds=torch.randn(ntok, 64, generator=g).to(torch.bfloat16),
                           pooled=torch.randn(32, generator=g).to(torch.bfloat16), cache_options={"cache_schema_version": 13})
            torch.save(payload, te)
            variants[ct] = te
        primary = variants.get("tags_nl") or variants["txt"]
        item = dict(meta, te_path=str(primary), lat_path=str(lat), image_file_signature=None, caption_file_signature=None,
                    caption_signature=None)
        if json_mode:
            item["caption_variants"] = {ct: {"te_path": str(variants[ct])} for ct in CAPTION_TYPES}
        files.append(item)
    files = files[::-1]                                  # the index order is arbitrary; the feed sorts it stably
    torch.save({"version": 13, "cache_options": {"cache_schema_version": 13}, "files": files}, cache / "dataset_index.pt")
    torch.save({"embeds": torch.randn(1, 77, 64, generator=g).to(torch.bfloat16), "pooled": torch.randn(1, 32, generator=g).to(torch.bfloat16)},
               cache / "null_embeds.pt")
    return cache
