"""Per-rank CPU placement (aozora_sdxl_training_amd/affinity.py): the pure partition logic and the idempotent bind."""
import os

from aozora_sdxl_training_amd import affinity


def test_cpulist_parse():
    assert affinity.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert affinity.parse_cpulist("") == []


def test_partition_follows_the_gpu_numa_node():
    # a two-socket node with SMT numbering: node 0 = 0-63 + 128-191, node 1 = 64-127 + 192-255; GPUs 0-3 on node 0, 4-7 on node 1
    node_cpus = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    gpu_nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    allowed = list(range(256))
    got = [affinity.partition(allowed, r, 8, gpu_nodes, node_cpus) for r in range(8)]
    assert all(len(g) == 32 for g in got)
    assert sorted(c for g in got for c in g) == allowed                          # a partition: disjoint, complete
    for r in range(8):
        assert set(got[r]) <= set(node_cpus[gpu_nodes[r]])                       # every rank on its GPU's socket
    # a contiguous split of the mask would have put rank 2 (cpus 64-95) on the other socket
    assert set(affinity.partition(allowed, 2, 8)) == set(range(64, 96))


def test_partition_fallbacks():
    allowed = list(range(16))
    # no NUMA information: even contiguous split by local rank
    assert affinity.partition(allowed, 1, 4) == [4, 5, 6, 7]
    assert affinity.partition(allowed, 3, 4, [-1, -1, -1, -1], {0: allowed}) == [12, 13, 14, 15]
    # a restricted mask (container cpuset) is respected: node CPUs outside it are not used
    assert affinity.partition([0, 1, 2, 3], 1, 2, [0, 0], {0: list(range(16))}) == [2, 3]
    # more ranks than CPUs: nobody gets an empty set
    assert affinity.partition([0, 1], 3, 8) == [0, 1]
    # one rank: the mask as it is
    assert affinity.partition([5, 6], 0, 1) == [5, 6]


def test_bind_rank_is_idempotent_and_caps_threads():
    import torch
    before = sorted(os.sched_getaffinity(0))
    affinity._DONE = None
    try:
        rec = affinity.bind_rank(local_rank=0, local_world=1, max_threads=2)
        assert rec["n_cpus"] == len(before) and rec["threads"] <= 2 and torch.get_num_threads() <= 2
        assert sorted(os.sched_getaffinity(0)) == before                          # one rank: the mask is left alone
        assert affinity.bind_rank(local_rank=5, local_world=8) is rec            # second call: no change
    finally:
        affinity._DONE = None
        torch.set_num_threads(max(1, min(8, len(before))))
