"""Stream concurrency probes for the two-stream executor and the exchange / copy streams.

HIP streams are multiplexed onto a handful of hardware queues (ROCclr: GPU_MAX_HW_QUEUES per priority, attached at a stream's
FIRST USE to the least-loaded queue; the queues in turn share the command processor's pipes), and nothing in the API says
which.  Measured on MI355X (tools/stream_probe.py, tools/ga1_probe.py):

  * two streams on one hardware queue run strictly one after the other (micro-step 166 ms instead of 138 ms);
  * of the first eight high-priority pool streams two are bad partners for a given normal-priority stream: independent kernels
    still overlap, but every cross-stream event hand-off (the executor's fork / join pattern) takes twice as long;
  * the 3rd, 4th ... high-priority stream a process uses can land where the full two-stream step thrashes (190-240 ms) although
    both probes below pass -- only the full step tells.

Policy that follows (train_step.py, dist.py): the streams are created once, in a fixed order and at normal priority (stream
priority measured no gain in a same-box A/B), every `TrainStep` of a UNet shares one data-gradient stream, and `check()` logs what the probes say
about the pairs that matter (bench.py prints the log).  `pick()` searches the pool for a stream that passes both probes
against a given set; it is a tool for experiments -- probing a candidate is its first use and attaches it to a queue, so a
search perturbs the mapping it inspects (picked m/v copy streams cost the bench 14 %).

Probes (`az_spin`: one idle wave for N microseconds): side by side (one spin on each stream: ~1.1x of one spin when they
overlap, 2x on one queue) and event ping-pong (12 round trips a -> b -> a of 50 us spins: 1.3-1.4x the pure spin time for a good
pair, 2.9x for a bad one).
"""
from __future__ import annotations

import time
from typing import List, Sequence

import torch

from ._lib import lib

PROBE_US, PROBE_ROUNDS, GOOD_RATIO = 50, 12, 2.0
log: List[str] = []          # what pick() decided, for diagnostics (bench.py prints it to stderr)


def _ptr(s: torch.cuda.Stream):
    import ctypes
    return ctypes.c_void_p(s.cuda_stream)


def pingpong_ratio(a: torch.cuda.Stream, b: torch.cuda.Stream) -> float:
    """Wall time of PROBE_ROUNDS event-chained round trips a -> b -> a of PROBE_US spins, over the pure spin time (best of 3,
    so a host hiccup cannot fake a bad pair)."""
    L = lib()
    best = float("inf")
    for _ in range(3):
        a.synchronize(); b.synchronize()
        t0 = time.perf_counter()
        for _ in range(PROBE_ROUNDS):
            L.call("az_spin", PROBE_US, _ptr(a))
            e = torch.cuda.Event(); e.record(a); b.wait_event(e)
            L.call("az_spin", PROBE_US, _ptr(b))
            e = torch.cuda.Event(); e.record(b); a.wait_event(e)
        a.synchronize(); b.synchronize()
        best = min(best, time.perf_counter() - t0)
        if best < 0.8 * GOOD_RATIO * 2e-6 * PROBE_ROUNDS * PROBE_US:
            break
    return best / (2e-6 * PROBE_ROUNDS * PROBE_US)


def spin_pair_ratio(a: torch.cuda.Stream, b: torch.cuda.Stream, us: int = 400) -> float:
    """Wall time of one `us` spin on each stream, issued together, over `us` (best of 3): ~1.1 side by side, ~2 when both
    streams sit on one hardware queue."""
    L = lib()
    best = float("inf")
    for _ in range(3):
        a.synchronize(); b.synchronize()
        t0 = time.perf_counter()
        L.call("az_spin", us, _ptr(a))
        L.call("az_spin", us, _ptr(b))
        a.synchronize(); b.synchronize()
        best = min(best, time.perf_counter() - t0)
        if best < 1.3e-6 * us:
            break
    return best / (1e-6 * us)


def overlaps(a: torch.cuda.Stream, b: torch.cuda.Stream) -> bool:
    """True when `a` and `b` are good partners: independent kernels run concurrently (not one hardware queue) and
    cross-stream event hand-offs between them are not slowed down (not one pipe)."""
    if a.cuda_stream == b.cuda_stream:
        return False
    return spin_pair_ratio(a, b) < 1.6 and pingpong_ratio(a, b) < GOOD_RATIO


def pick(device, priority: int = 0, beside: Sequence[torch.cuda.Stream] = (), tries: int = 8, what: str = "stream") -> torch.cuda.Stream:
    """A pool stream of `priority` that runs concurrently with every stream in `beside` (most important first).  If no
    candidate overlaps with all of them, the one overlapping with the longest prefix of `beside` is returned."""
    device = torch.device(device)
    best, best_score = None, -1
    with torch.cuda.device(device):
        lib().call("az_spin", 1, _ptr(torch.cuda.current_stream(device)))     # module load outside the timed probes
        torch.cuda.current_stream(device).synchronize()
        seen = set()
        for k in range(max(1, tries)):
            s = torch.cuda.Stream(device=device, priority=priority)
            if s.cuda_stream in seen:
                break
            seen.add(s.cuda_stream)
            score = 0
            for other in beside:
                if not overlaps(s, other):
                    break
                score += 1
            if score > best_score:
                best, best_score = s, score
            if score == len(beside):
                if k:
                    log.append(f"{what}: candidate {k} taken ({k} earlier one(s) serialised with a stream it must run beside)")
                return s
    log.append(f"{what}: no candidate overlaps all {len(beside)} streams; best overlaps the first {best_score}")
    return best


def check(a: torch.cuda.Stream, b: torch.cuda.Stream, what: str, if_bad: str = "BAD PAIR (expect a slow step)") -> bool:
    """Probe a pair the executor relies on and log the verdict (never changes the streams: probing a stream for the first
    time attaches it to a hardware queue, so searching perturbs the very mapping it inspects)."""
    sp, pp = spin_pair_ratio(a, b), pingpong_ratio(a, b)
    ok = sp < 1.6 and pp < GOOD_RATIO
    log.append(f"{what}: side-by-side {sp:.2f}x, event ping-pong {pp:.2f}x -> {'ok' if ok else if_bad}")
    return ok


_host_link = {}


def host_link_streams(device):
    """The two streams (H2D, D2H) that carry the Raven / Titan state over the host link, one pair per device and process, created
    AND USED at the first call: each makes one pinned 64-MiB copy in its direction, which binds it to an SDMA engine.

    Call this BEFORE torch.distributed creates its RCCL communicator (bench.py and trainer.main do; ShardedRaven / ShardedTitan
    call it again and get the same pair).  Measured (tools/iter_timeline.py, profiles/r04_host_link_and_rccl.txt): a copy stream whose
    first copy comes after `init_process_group(backend="nccl", device_id=...)` no longer gets the engine -- the 182-ms H2D of m / v
    then runs (partly) as blit kernels whose reads of host memory sit in the L2's request queues, and the two micro-steps beside it
    take 137-141 ms instead of 116 (with every copy forced onto blit kernels, HSA_ENABLE_SDMA=0: 299 ms); at eight ranks the one
    micro-step of an iteration would be that one.  A library-owned copy kernel of a few workgroups is no way out: it fills the link
    at 8 workgroups and stalls the step just the same (123-300 ms per micro-step, chunked or not)."""
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key in _host_link:
        return _host_link[key]
    late = False
    try:
        import torch.distributed as dist
        late = dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"
    except Exception:
        late = False
    with torch.cuda.device(device):
        h2d, d2h = torch.cuda.Stream(device), torch.cuda.Stream(device)
        hb = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
        db = torch.empty(64 << 20, dtype=torch.uint8, device=device)
        with torch.cuda.stream(h2d):
            db.copy_(hb, non_blocking=True)
        h2d.synchronize()
        with torch.cuda.stream(d2h):
            hb.copy_(db, non_blocking=True)
        d2h.synchronize()
    log.append("host-link streams: first copies made " + ("AFTER the RCCL communicator was created (expect the m / v H2D to slow the micro-steps beside it)"
                                                          if late else "before any RCCL communicator exists -> ok"))
    _host_link[key] = (h2d, d2h)
    return _host_link[key]
