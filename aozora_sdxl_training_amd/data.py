"""Data feed of the training step (SURVEY.md 8f rows f1 + f2): everything between the offline VAE / CLIP cache on disk
and the `batch` dict the loop body consumes (train.py:2709-2741).  Host-side Python.

PORT NOTICE: what is a bit-exact CONTRACT with the reference's caches and resumes follows the reference (Hysocs/Aozora_SDXL_Training
train.py and training_utils/caching/cache.py, Apache-2.0; see the NOTICE file at the repository root) statement order for
statement order where the order decides the result: the draws of every seeded generator (per-sample sha256 key -> caption variant,
dropout, conditioning scale; per-epoch torch.randperm; the bin-spread PCG64 stream; BucketBatchSampler's interleave), the on-disk
schema keys and the stable item order.  Everything around those draws -- path helpers, caption-share table, bin lookup, schedule
assembly, the dataset item's scaffolding -- is this module's own (tests/test_data_feed.py replays synthetic caches against the
imported reference's classes either way).  Mirrors

  cache index / path helpers              training_utils/caching/cache.py:9-246          (f2)
  ImageTextLatentDataset                  train.py:1992-2160   -> CachedLatentDataset    (f1)
  BucketBatchSampler, Precomputed...      train.py:461-563     -> BucketBatchSampler, PrecomputedBatchSampler
  image / batch schedules (epoch, spread) train.py:688-882     -> image_schedule(), batch_schedule()
  pack_sdxl_sample_schedule, collate      train.py:2213-2254   -> pack_schedule(), collate()

Everything here is pinned against the reference itself: tests/golden/make_golden_data.py runs the reference's classes over
synthetic caches (tests/golden/synth_cache.py) and tests/test_data_feed.py replays the same caches through this module --
item order, sampler batches, both schedules, packed ids, per-sample caption-variant choice / null-conditioning dropout /
conditioning-scale lerp (keyed by sha256(seed, sample position)), dropped NaN latents and the collated batch all match.

The consumer is unchanged: `torch.utils.data.DataLoader(dataset, batch_sampler=PrecomputedBatchSampler(...),
collate_fn=collate)` as in train.py:2655-2658; for data parallel runs rank r feeds rows [r*b, (r+1)*b) of every global
batch (`shard_batch`), matching TimestepSampler.sample_shard.
"""
from __future__ import annotations

import hashlib
import itertools
import math
import random
import re
from collections import defaultdict
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

# ---- cache layout (cache.py:9-41) -----------------------------------------------------------------------------------
CAPTION_JSON_TYPES = ("tags", "nl", "tags_nl", "nl_tags")
CAPTION_JSON_PRIMARY_TYPE = "tags_nl"
CACHE_INDEX_NAME = "dataset_index.pt"
CLIP_CHUNK_TOKEN_COUNT = 77
_JSON_SUFFIX = re.compile(r"_json_(tags|nl|tags_nl|nl_tags)$")
_MB_SUFFIX = re.compile(r"_mb\d+$")


def cache_folder_name(is_rectified_flow: bool) -> str:
    """train.py:2002: the two training modes keep separate (identically formatted) caches."""
    return ".precomputed_embeddings_cache_rf" if is_rectified_flow else ".precomputed_embeddings_cache_standard_sdxl"


def load_cache_index(where):
    """The cache's file list: `where` is the cache directory or the index file itself (cache.py:83-90 reads the same file)."""
    where = Path(where)
    index_file = where if where.is_file() else where / CACHE_INDEX_NAME
    return torch.load(index_file, map_location="cpu", weights_only=False)


def caption_source_type(config_or_value=None) -> str:
    v = config_or_value
    if v is not None and not isinstance(v, str):
        v = getattr(v, "CAPTION_SOURCE_TYPE", "txt")
    return "json" if str(v or "txt").strip().lower() == "json" else "txt"


# caption variant -> (preset key, default share in percent); the preset keys and defaults are the reference's (train.py:86-96)
_CAPTION_SHARE_KEYS = {"tags": ("CAPTION_TAGS_PERCENT", 40), "nl": ("CAPTION_NL_PERCENT", 10),
                       "tags_nl": ("CAPTION_TAGS_NL_PERCENT", 25), "nl_tags": ("CAPTION_NL_TAGS_PERCENT", 25)}


def json_caption_weights(config) -> Dict[str, int]:
    """Integer share of every JSON caption variant, never negative; all-zero shares fall back to the primary variant alone."""
    shares = {}
    for variant in CAPTION_JSON_TYPES:
        key, default = _CAPTION_SHARE_KEYS[variant]
        shares[variant] = max(0, int(getattr(config, key, default) or 0))
    if not any(shares.values()):
        shares[CAPTION_JSON_PRIMARY_TYPE] = 100
    return shares


def text_conditioning_scale_range(config):
    """train.py:1227-1236."""
    if not bool(getattr(config, "TEXT_CONDITIONING_SCALE_ENABLED", False)):
        return 1.0, 1.0
    lo = min(max(float(getattr(config, "TEXT_CONDITIONING_SCALE_MIN", 1.0)), 0.0), 1.0)
    hi = min(max(float(getattr(config, "TEXT_CONDITIONING_SCALE_MAX", 1.0)), 0.0), 2.0)
    return (hi, lo) if lo > hi else (lo, hi)


def stable_item_key(item):
    """cache.py:113-121: order of cached items independent of filesystem traversal."""
    norm = lambda s: str(s).replace("\\", "/").casefold()
    return (norm(item.get("relative_path", item.get("image_key", ""))), int(item.get("bucket_variant_index", 0) or 0),
            tuple(item.get("target_size", (0, 0))), norm(item.get("lat_path", item.get("te_path", ""))))


def item_stem_from_te_path(path) -> Optional[str]:
    name = Path(path).name
    return _JSON_SUFFIX.sub("", name[:-len("_te.pt")]) if name.endswith("_te.pt") else None


def base_stem_from_cache_path(path) -> Optional[str]:
    name = Path(path).name
    if name.endswith("_te.pt"):
        return _MB_SUFFIX.sub("", item_stem_from_te_path(path))
    if name.endswith("_lat.pt"):
        return _MB_SUFFIX.sub("", name[:-len("_lat.pt")])
    return None


def lat_path_for_te_path(te_path) -> Path:
    te_path = Path(te_path)
    stem = item_stem_from_te_path(te_path)
    return Path(str(te_path).replace("_te.pt", "_lat.pt")) if stem is None else te_path.with_name(f"{stem}_lat.pt")


def choose_caption_variant(rng: random.Random, weights) -> str:
    """cache.py:209-220: one rng.uniform draw over the cumulative integer weights (in CAPTION_JSON_TYPES order)."""
    w = [max(0, int(weights.get(k, 0) or 0)) for k in CAPTION_JSON_TYPES]
    total = sum(w)
    if total <= 0:
        return CAPTION_JSON_PRIMARY_TYPE
    roll, upto = rng.uniform(0, total), 0
    for k, wk in zip(CAPTION_JSON_TYPES, w):
        upto += wk
        if roll <= upto:
            return k
    return CAPTION_JSON_PRIMARY_TYPE


def select_te_path(item, rng, weights, json_mode: bool):
    """Text-embedding file of one sample.  Plain caches have one per item; JSON-caption caches carry one per caption variant and
    the sample's own generator draws which (ONE draw, cache.py:231-240 -- the draw is part of the bit-exact contract; a variant
    without a file falls back to the primary one, then to any)."""
    variants = item.get("caption_variants") if json_mode else None
    if not isinstance(variants, dict):
        return item.get("te_path")
    drawn = choose_caption_variant(rng, {name: weights.get(name, 0) for name in variants})
    for candidate in (variants.get(drawn), variants.get(CAPTION_JSON_PRIMARY_TYPE), next(iter(variants.values()), None)):
        if candidate:
            return candidate["te_path"] if isinstance(candidate, dict) and candidate.get("te_path") else item.get("te_path")
    return item.get("te_path")


# ---- dataset (train.py:1992-2160) -----------------------------------------------------------------------------------
SAMPLE_INDEX_BITS = 32
_MASK = (1 << SAMPLE_INDEX_BITS) - 1


def pack_sample_index(dataset_index: int, sample_index: int) -> int:
    dataset_index, sample_index = int(dataset_index), int(sample_index)
    if not 0 <= dataset_index <= _MASK:
        raise ValueError(f"Dataset index is too large to pack deterministically: {dataset_index}")
    return (sample_index << SAMPLE_INDEX_BITS) | dataset_index


def unpack_sample_index(packed: int):
    packed = int(packed)
    return packed & _MASK, packed >> SAMPLE_INDEX_BITS


class CachedLatentDataset(torch.utils.data.Dataset):
    """Index = packed (dataset position, absolute sample position); the sample position seeds all per-sample randomness,
    so a resumed or re-sharded run draws the same caption variant / dropout / scale for the same training sample."""

    pack_sample_index = staticmethod(pack_sample_index)
    unpack_sample_index = staticmethod(unpack_sample_index)

    def __init__(self, config):
        self.seed = config.SEED if config.SEED else 42
        self.json_caption_mode = caption_source_type(config) == "json"
        self.caption_weights = json_caption_weights(config)
        folder = cache_folder_name(bool(config.is_rectified_flow))
        items = []
        for ds in config.INSTANCE_DATASETS:
            cache_dir = Path(ds["path"]) / folder
            if not (cache_dir / CACHE_INDEX_NAME).exists():
                print(f"WARNING: Index missing at {cache_dir}. Please re-run caching!")
                continue
            ordered = sorted(load_cache_index(cache_dir)["files"], key=stable_item_key)
            items.extend(ordered * int(ds.get("repeats", 1)))
        if not items:
            raise ValueError("No cached files found.")
        random.Random(self.seed).shuffle(items)
        self.items = items
        self.bucket_keys = [tuple(it["target_size"]) for it in items]
        self.cond_scale_min, self.cond_scale_max = text_conditioning_scale_range(config)
        self.cond_scale_enabled = self.cond_scale_min < 1.0 or self.cond_scale_max > 1.0
        self.dropout_prob = (min(max(float(getattr(config, "UNCONDITIONAL_DROPOUT_CHANCE", 0.0)), 0.0), 1.0)
                             if getattr(config, "UNCONDITIONAL_DROPOUT", False) else 0.0)
        self.null_embeds = self.null_pooled = None
        if self.dropout_prob > 0 or self.cond_scale_enabled:
            try:
                nd = torch.load(Path(config.INSTANCE_DATASETS[0]["path"]) / folder / "null_embeds.pt", map_location="cpu", weights_only=True)
                self.null_embeds = nd["embeds"].squeeze(0) if nd["embeds"].dim() == 3 else nd["embeds"]
                self.null_pooled = nd["pooled"].squeeze(0) if nd["pooled"].dim() == 2 else nd["pooled"]
            except Exception:
                self.dropout_prob, self.cond_scale_enabled = 0.0, False

    def __len__(self):
        return len(self.items)

    def _rng_for_sample(self, dataset_index, sample_index) -> random.Random:
        digest = hashlib.sha256(f"{self.seed}:sdxl-sample:{int(sample_index)}:{int(dataset_index)}".encode("utf-8")).digest()
        return random.Random(int.from_bytes(digest[:8], "little"))

    # -- null conditioning of a different token length (caption chunking): train.py:2066-2110
    def _null_of_length(self, n, dtype):
        ne = self.null_embeds
        if ne is None:
            return None
        have = ne.shape[0]
        if n <= have:
            return ne[:n].to(dtype=dtype)
        chunk = CLIP_CHUNK_TOKEN_COUNT if have >= CLIP_CHUNK_TOKEN_COUNT else have
        if chunk <= 0 or have % chunk != 0:
            return torch.cat([ne, ne[-1:].expand(n - have, -1)], dim=0).to(dtype=dtype)
        tail = ne[-chunk:]
        whole, part = divmod(n - have, chunk)
        pieces = [ne] + ([tail.repeat(whole, 1)] if whole else []) + ([tail[:part]] if part else [])
        return torch.cat(pieces, dim=0).to(dtype=dtype)

    def _aligned(self, embeds):
        ne = self.null_embeds
        if ne is None or embeds.shape == ne.shape or embeds.dim() != 2 or ne.dim() != 2 or embeds.shape[1] != ne.shape[1]:
            return embeds, ne
        if embeds.shape[0] < ne.shape[0]:
            embeds = torch.cat([embeds, self._null_of_length(ne.shape[0], embeds.dtype)[embeds.shape[0]:ne.shape[0]]], dim=0)
        elif embeds.shape[0] > ne.shape[0]:
            ne = self._null_of_length(embeds.shape[0], ne.dtype)
        return embeds, ne

    @staticmethod
    def _unbatched(t, batched_rank):
        return t.squeeze(0) if t.dim() == batched_rank else t

    def _read_sample(self, meta, te_path):
        """Both cache files of one sample -> (latents, embeds [L, D], pooled [P]); None when the latents are not finite."""
        te = torch.load(te_path, map_location="cpu", weights_only=True)
        lat = torch.load(meta["lat_path"], map_location="cpu", weights_only=True)
        latents = lat["latents"] if isinstance(lat, dict) else lat
        if not bool(torch.isfinite(latents).all()):
            return None
        return latents, self._unbatched(te["embeds"], 3), self._unbatched(te["pooled"], 2)

    def _condition(self, rng, embeds, pooled):
        """Per-sample conditioning changes, in the reference's draw order (train.py:2141-2152): the dropout draw first -- taken
        only when dropout is on -- then, for a kept sample, the scale draw when scaling is on."""
        if self.dropout_prob > 0 and rng.random() < self.dropout_prob:
            return self._aligned(embeds)[1], self.null_pooled                      # unconditional sample
        if self.cond_scale_enabled:
            s = rng.uniform(self.cond_scale_min, self.cond_scale_max)
            e, ne = self._aligned(embeds)
            return ne + (e - ne) * s, self.null_pooled + (pooled - self.null_pooled) * s      # lerp towards / past the null conditioning
        return embeds, pooled

    def __getitem__(self, packed):
        try:
            position, sample = unpack_sample_index(packed)
            meta = self.items[position]
            rng = self._rng_for_sample(position, sample)
            te_path = select_te_path(meta, rng, self.caption_weights, self.json_caption_mode)      # (first use of the sample's generator)
            loaded = self._read_sample(meta, te_path)
            if loaded is None:
                return None
            latents, embeds, pooled = loaded
            embeds, pooled = self._condition(rng, embeds, pooled)
            original = meta["original_size"]
            return dict(latents=latents, embeds=embeds, pooled=pooled, original_sizes=original,
                        scaled_sizes=meta.get("scaled_size", original), target_sizes=meta["target_size"],
                        crop_coords=meta.get("crop_coords", (0, 0)), latent_path=te_path,
                        image_key=meta.get("relative_path", meta["lat_path"]))
        except Exception as e:
            print(f"[DATASET] Failed to load item {packed}: {e}")
            return None


def collate(batch):
    """train.py:2213-2221: drop failed samples, stack tensors, keep the rest as lists."""
    batch = [b for b in batch if b]
    if not batch:
        return {}
    return {k: (torch.stack([b[k] for b in batch]) if isinstance(batch[0][k], torch.Tensor) and k != "original_image" else [b[k] for b in batch])
            for k in batch[0]}


def shard_rows(n: int, rank: int, world: int) -> slice:
    """Rows of a global batch of n samples that rank `rank` processes: contiguous shares whose sizes differ by at most
    one (a ragged batch -- bucket remainder, dropped sample -- gives the first n % world ranks one row more)."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return slice(lo, lo + q + (1 if rank < r else 0))


def shard_batch(batch: Dict, rank: int, world: int, even: bool = True) -> Dict:
    """Data parallel: this rank's rows of a collated GLOBAL batch (same slicing as TimestepSampler.sample_shard when the
    batch divides evenly; `even=False` allows ragged shares, see shard_rows)."""
    n = len(batch["target_sizes"])
    if even and n % world:
        raise ValueError(f"global batch {n} is not divisible by world size {world}")
    rows = shard_rows(n, rank, world)
    return {k: v[rows] for k, v in batch.items()}


# ---- samplers (train.py:461-563) ------------------------------------------------------------------------------------
class BucketBatchSampler(torch.utils.data.Sampler):
    """One resolution bucket per batch.  Per epoch: a seeded permutation is split into per-bucket chunks; with shuffle
    the chunks of a bucket are permuted and buckets are interleaved largest-backlog-first, never repeating the previous
    bucket when another one is available (ties broken by the same generator)."""

    def __init__(self, dataset, batch_size, seed, shuffle=True):
        self.dataset, self.batch_size, self.seed, self.shuffle = dataset, batch_size, seed, shuffle
        self.epoch = 0
        self.start_batch_index = 0
        self.total_images = len(dataset)

    def set_epoch(self, epoch):
        self.epoch = epoch

    def set_start_batch_index(self, batch_index):
        self.start_batch_index = max(0, int(batch_index or 0))

    def _epoch_batches(self) -> List[List[int]]:
        g = torch.Generator()
        g.manual_seed(self.seed + self.epoch)
        order = torch.randperm(self.total_images, generator=g).tolist()
        if self.batch_size == 1:
            return [[i] for i in order]
        per_bucket = defaultdict(list)
        for i in order:
            per_bucket[self.dataset.bucket_keys[i]].append(i)
        pending = {}
        for key in sorted(per_bucket):
            idx = per_bucket[key]
            chunks = [idx[a:a + self.batch_size] for a in range(0, len(idx), self.batch_size)]
            if self.shuffle and len(chunks) > 1:
                chunks = [chunks[j] for j in torch.randperm(len(chunks), generator=g).tolist()]
            pending[key] = chunks
        if not self.shuffle:
            return [b for key in sorted(pending) for b in pending[key]]
        out, last = [], None
        while pending:
            pool = [k for k in pending if k != last] or list(pending)
            most = max(len(pending[k]) for k in pool)
            top = [k for k in pool if len(pending[k]) == most]
            key = top[torch.randint(len(top), (1,), generator=g).item()]
            out.append(pending[key].pop(0))
            last = key
            if not pending[key]:
                del pending[key]
        return out

    def __iter__(self):
        batches = self._epoch_batches()
        if self.start_batch_index > 0:
            batches = batches[self.start_batch_index:]
            self.start_batch_index = 0
        self.epoch += 1
        yield from batches

    def __len__(self):
        return math.ceil(self.total_images / self.batch_size)


class PrecomputedBatchSampler(torch.utils.data.Sampler):
    """train.py:540-563: walks a precomputed (packed) batch schedule; `start_step` resumes mid-run."""

    def __init__(self, image_batches, seed, start_step=0):
        self.image_batches, self.seed = image_batches, seed
        self.start_step = max(0, int(start_step or 0))
        self.epoch = 0

    def __iter__(self):
        for step in range(self.start_step, len(self.image_batches)):
            self.epoch = step + 1
            b = self.image_batches[step]
            yield [int(i) for i in (b.tolist() if isinstance(b, np.ndarray) else b)]

    def __len__(self):
        return max(0, len(self.image_batches) - self.start_step)

    def set_epoch(self, epoch):
        self.epoch = int(epoch or 0)

    def set_start_batch_index(self, batch_index):
        self.start_step = max(0, int(batch_index or 0))


# ---- schedules (train.py:566-574, 688-882) --------------------------------------------------------------------------
def timestep_bin_ids(timesteps, bin_ranges) -> np.ndarray:
    """Index of the FIRST half-open range [lo, hi) holding each timestep (0 when none does), for all timesteps at once."""
    t = np.asarray([int(x) for x in timesteps], dtype=np.int64)
    if t.size == 0 or len(bin_ranges) == 0:
        return np.zeros(t.size, dtype=np.int32)
    lo = np.asarray([r[0] for r in bin_ranges], dtype=np.int64)
    hi = np.asarray([r[1] for r in bin_ranges], dtype=np.int64)
    inside = (t[:, None] >= lo[None, :]) & (t[:, None] < hi[None, :])
    return np.where(inside.any(axis=1), inside.argmax(axis=1), 0).astype(np.int32)


def _epoch_permutation(total_images, seed, epoch) -> np.ndarray:
    """The image order of one epoch: torch.randperm under a generator seeded with seed + epoch (the bit-exact part, train.py:688-700)."""
    g = torch.Generator()
    g.manual_seed(seed + epoch)
    return torch.randperm(total_images, generator=g).numpy().astype(np.uint32, copy=False)


def _epoch_image_schedule(total_images, total_steps, seed) -> np.ndarray:
    """One image per step: whole epochs back to back, the last one cut at total_steps."""
    if total_steps <= 0 or total_images <= 0:
        return np.empty(0, dtype=np.uint32)
    epochs = -(-total_steps // total_images)
    return np.concatenate([_epoch_permutation(total_images, seed, e) for e in range(epochs)])[:total_steps]


class _BinSpread:
    """Shared state of the 'spread' schedules: each image remembers the last `depth` timestep bins it was paired with and is
    steered away from bins it has seen recently; within an epoch every image is used at most once."""

    def __init__(self, total_images, bin_count, samples):
        self.n = total_images
        self.depth = max(1, min(bin_count, math.ceil(samples / total_images)))
        wide = bin_count >= 255
        self.recent = np.full((total_images, self.depth), 65535 if wide else 255, dtype=np.uint16 if wide else np.uint8)
        self.cursor = np.zeros(total_images, dtype=np.uint16)

    def start_epoch(self, seed, epoch):
        self.free = np.ones(self.n, dtype=np.bool_)
        self.queues, self.pos = {}, {}
        self.rng = np.random.Generator(np.random.PCG64(seed + 104729 + epoch))

    def pick(self, queue_key, bin_id, make_queue, candidates_left):
        """Next unused image of `queue_key`'s queue that has not met `bin_id` recently; else the least-penalised free one."""
        q = self.queues.get(queue_key)
        if q is None:
            q = self.queues[queue_key] = make_queue(self.rng)
            self.pos[queue_key] = 0
        p, chosen = self.pos[queue_key], None
        while p < len(q):
            c = int(q[p])
            p += 1
            if self.free[c] and not np.any(self.recent[c] == bin_id):
                chosen = c
                break
        self.pos[queue_key] = p
        if chosen is None:
            left = candidates_left(self.free)
            if left.size == 0:
                return None
            pen = np.count_nonzero(self.recent[left] == bin_id, axis=1)
            best = left[pen == pen.min()]
            chosen = int(best[int(self.rng.integers(0, len(best)))])
        self.free[chosen] = False
        self.recent[chosen, int(self.cursor[chosen] % self.depth)] = bin_id
        self.cursor[chosen] = (self.cursor[chosen] + 1) % self.depth
        return chosen


def _spread_image_schedule(total_images, total_steps, seed, bin_ids, bin_count) -> np.ndarray:
    if total_images <= 0 or total_steps <= 0:
        return np.empty(0, dtype=np.uint32)
    if bin_count <= 1:
        return _epoch_image_schedule(total_images, total_steps, seed)
    st = _BinSpread(total_images, bin_count, total_steps)
    out = np.empty(total_steps, dtype=np.uint32)
    done = epoch = 0
    while done < total_steps:
        span = min(total_images, total_steps - done)
        st.start_epoch(seed, epoch)
        for k in range(span):
            b = int(bin_ids[done + k])
            c = st.pick(b, b, lambda rng: rng.permutation(total_images).astype(np.uint32, copy=False), lambda free: np.flatnonzero(free))
            if c is None:
                break
            out[done + k] = c
        done += span
        epoch += 1
    return out


def image_schedule(total_images, total_steps, seed, timesteps, bin_ranges, force_spread) -> np.ndarray:
    """train.py:765-774 (batch size 1 view: one image per step)."""
    if not force_spread:
        return _epoch_image_schedule(total_images, total_steps, seed)
    return _spread_image_schedule(total_images, total_steps, seed, timestep_bin_ids(timesteps, bin_ranges), len(bin_ranges))


def _epoch_batch_schedule(dataset, total_steps, batch_size, seed) -> List[List[int]]:
    """The first total_steps batches of the endless stream 'epoch 0's bucket batches, epoch 1's, ...' (train.py:777-790)."""
    if len(dataset) == 0:
        return []

    def stream():
        for epoch in itertools.count():
            sampler = BucketBatchSampler(dataset, batch_size, seed, shuffle=True)
            sampler.set_epoch(epoch)
            yield from sampler
    return [[int(i) for i in batch] for batch in itertools.islice(stream(), max(0, total_steps))]


def _spread_batch_schedule(dataset, total_steps, batch_size, seed, timesteps, bin_ranges) -> List[List[int]]:
    n = len(dataset)
    if n <= 0 or total_steps <= 0:
        return []
    if batch_size == 1:
        return [[int(i)] for i in image_schedule(n, total_steps, seed, timesteps, bin_ranges, True).tolist()]
    bin_ids = timestep_bin_ids(timesteps, bin_ranges)
    st = _BinSpread(n, max(1, len(bin_ranges)), min(len(timesteps), total_steps * batch_size))
    members = defaultdict(list)
    for i, key in enumerate(dataset.bucket_keys):
        members[key].append(i)
    out, used, epoch = [], 0, 0
    while len(out) < total_steps:
        base = BucketBatchSampler(dataset, batch_size, seed, shuffle=True)     # supplies the bucket order and batch sizes
        base.set_epoch(epoch)
        st.start_epoch(seed, epoch)
        for proto in base:
            if len(out) >= total_steps:
                break
            bucket = dataset.bucket_keys[proto[0]]
            pool = members[bucket]

            def fresh_queue(rng, pool=pool):
                q = np.array(pool, dtype=np.uint32)
                rng.shuffle(q)
                return q

            chosen = []
            for j in range(len(proto)):
                if used + j >= len(bin_ids):
                    break
                b = int(bin_ids[used + j])
                c = st.pick((bucket, b), b, fresh_queue, lambda free, pool=pool: np.array([i for i in pool if free[i]], dtype=np.int64))
                if c is None:
                    break
                chosen.append(c)
            if chosen:
                out.append(chosen)
                used += len(chosen)
            if used >= len(bin_ids):
                break
        epoch += 1
    return out


def batch_schedule(dataset, total_steps, batch_size, seed, timesteps, bin_ranges, force_spread) -> List[List[int]]:
    """train.py:879-882: the run's whole batch schedule (dataset positions), fixed before the first step."""
    if not force_spread:
        return _epoch_batch_schedule(dataset, total_steps, batch_size, seed)
    return _spread_batch_schedule(dataset, total_steps, batch_size, seed, timesteps, bin_ranges)


def pack_schedule(schedule: Sequence[Sequence[int]], batch_size) -> List[List[int]]:
    """train.py:2245-2254: attach the absolute sample position (batch_index * batch_size + row) to every entry."""
    bs = max(1, int(batch_size or 1))
    return [[pack_sample_index(d, bi * bs + li) for li, d in enumerate(batch)] for bi, batch in enumerate(schedule)]
