"""Per-rank CPU placement for one-process-per-GPU runs (the reference is single-device, train.py:2551).

Each rank of an N-GPU node pins 1/N of the optimizer state in host memory (dist.ShardedRaven: m / v shards) and streams it over
its GPU's host link every optimizer step; its loop also runs a handful of small CPU tensor ops per micro-step.  Left alone, the
eight ranks of a node each start a torch intra-op pool as wide as the machine (8 x 128-256 threads) and first-touch their
pinned buffers on whatever socket the kernel happens to schedule them on -- for half of the GPUs the wrong one.

`bind_rank()` runs ONCE per process, before the pinned allocations and before torch's intra-op pool exists:

  * the rank's CPUs = the CPUs of the NUMA node its GPU hangs off (sysfs: /sys/bus/pci/devices/<bdf>/numa_node and
    /sys/devices/system/node/node<k>/cpulist), intersected with the process' current affinity mask and divided evenly among the
    local ranks whose GPUs share that node; with no NUMA information (numa_node = -1, containers without sysfs) an even
    contiguous split of the current mask by LOCAL_RANK;
  * `os.sched_setaffinity` to that set -- no numactl / taskset wrapper: a launcher hop between a profiler and the program is
    the exec the GPU boxes forbid;
  * torch intra-op threads capped (default 8, never more than the rank's CPUs).

Memory follows the CPUs: Linux' default policy is local allocation, so buffers first touched after the bind (hipHostMalloc
touches while pinning) land on the rank's node.  The function never raises: a box that refuses the call keeps its old mask and
the returned record says so.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

_DONE: Optional[dict] = None


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    out: List[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def partition(allowed: Sequence[int], local_rank: int, local_world: int, gpu_nodes: Optional[Sequence[int]] = None,
              node_cpus: Optional[Dict[int, Sequence[int]]] = None) -> List[int]:
    """The CPUs of `local_rank` (pure function; tests/test_affinity_cpu.py).

    allowed: the process' current affinity mask.  gpu_nodes[r]: NUMA node of local rank r's GPU (-1 / None: unknown).
    node_cpus[k]: CPUs of node k.  Ranks whose GPUs share a node divide that node's allowed CPUs evenly, in rank order; a
    rank whose node is unknown, or whose share would be empty, gets its slice of an even contiguous split of `allowed`."""
    allowed = sorted(set(int(c) for c in allowed))
    if local_world <= 1 or not allowed:
        return list(allowed)

    def contiguous() -> List[int]:
        per = len(allowed) // local_world
        if per == 0:
            return list(allowed)
        return allowed[local_rank * per:(local_rank + 1) * per]

    if not gpu_nodes or not node_cpus or len(gpu_nodes) != local_world:
        return contiguous()
    node = gpu_nodes[local_rank]
    if node is None or node < 0 or node not in node_cpus:
        return contiguous()
    peers = [r for r in range(local_world) if gpu_nodes[r] == node]
    mine = [c for c in sorted(set(node_cpus[node])) if c in set(allowed)]
    per = len(mine) // len(peers)
    if per == 0:
        return contiguous()
    k = peers.index(local_rank)
    return mine[k * per:(k + 1) * per]


def _gpu_numa_nodes(local_world: int) -> Optional[List[int]]:
    """NUMA node of every local GPU from sysfs (device properties only: no context is created on the other ranks' devices)."""
    try:
        import torch
        nodes = []
        for i in range(local_world):
            p = torch.cuda.get_device_properties(i)
            bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
            with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
                nodes.append(int(f.read().strip()))
        return nodes
    except Exception:
        return None


def _node_cpus() -> Optional[Dict[int, List[int]]]:
    try:
        base = "/sys/devices/system/node"
        out = {}
        for name in os.listdir(base):
            if name.startswith("node") and name[4:].isdigit():
                with open(os.path.join(base, name, "cpulist")) as f:
                    out[int(name[4:])] = parse_cpulist(f.read())
        return out or None
    except Exception:
        return None


def bind_rank(local_rank: Optional[int] = None, local_world: Optional[int] = None, max_threads: int = 8) -> dict:
    """Bind this process to its rank's CPUs and cap torch's intra-op threads (see the module text).  Idempotent; returns a
    record {cpus, n_cpus, numa_node, threads, how} for the bench line / the log."""
    global _DONE
    if _DONE is not None:
        return _DONE
    import torch
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if local_rank is None else int(local_rank)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))) if local_world is None else int(local_world)
    rec = dict(local_rank=local_rank, local_world=local_world, numa_node=None, how="unchanged")
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = list(range(os.cpu_count() or 1))
    cpus = allowed
    if local_world > 1:
        nodes, ncpus = _gpu_numa_nodes(local_world), _node_cpus()
        cpus = partition(allowed, local_rank, local_world, nodes, ncpus) or allowed
        known = bool(nodes and ncpus and len(nodes) == local_world and nodes[local_rank] is not None and nodes[local_rank] >= 0)
        rec["numa_node"] = nodes[local_rank] if known else None
        rec["how"] = "numa node of the GPU" if known else "even split of the affinity mask (no NUMA information)"
        try:
            os.sched_setaffinity(0, cpus)
        except (AttributeError, OSError, ValueError) as e:
            rec["how"] = f"sched_setaffinity refused ({e!r}): mask unchanged"
            cpus = allowed
    threads = max(1, min(int(max_threads), len(cpus)))
    if torch.get_num_threads() > threads:
        torch.set_num_threads(threads)
    rec.update(cpus=f"{cpus[0]}-{cpus[-1]}" if cpus and cpus == list(range(cpus[0], cpus[-1] + 1)) else ",".join(map(str, cpus[:64])),
               n_cpus=len(cpus), threads=torch.get_num_threads())
    _DONE = rec
    return rec
