"""Multi-GPU host logic: utterances are independent, so N GPUs = N replicas of the voice, one process per GPU.

The only collective is a ONE-SHOT broadcast of the packed weight blob from rank 0 (RCCL over xGMI when the backend is
"nccl"; the same code runs over gloo on CPUs for tests).  No per-step collectives: ranks synthesise disjoint shards of
the utterance batch and return their own waveforms.
"""
import numpy as np

FIXTURE_IDS = [1, 20, 0, 120, 0, 61, 0, 24, 0, 59, 0, 100, 0, 2]  # bench/fixtures/test_summary.json:8


def batch32_factors(seed=1234):
    """BASELINE configs[3]: 32 mixed-length utterances, factors [1,2,3,4,6,8,12,16]×4, order shuffled (seeded)."""
    f = [1, 2, 3, 4, 6, 8, 12, 16] * 4
    rng = np.random.RandomState(seed)  # MT19937 stream: stable across numpy versions
    rng.shuffle(f)
    return f


def shard_utterances(costs, world_size):
    """Longest-processing-time-first greedy partition: returns per-rank lists of utterance indices.

    cost ∝ frame count ∝ phoneme count (F = 3·T with pinned durations).  Deterministic (ties by index)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0] * world_size
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += costs[i]
    return shards


def broadcast_blob(blob, src=0):
    """Broadcast a 1-D float32 torch tensor (CPU for gloo, CUDA for nccl/RCCL) in place from `src`; returns it."""
    import torch.distributed as dist
    dist.broadcast(blob, src=src)
    return blob


def max_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


_meta_group = None


def gather_rows(row):
    """Every rank's small metadata dict on every rank (list indexed by rank). Goes over a gloo side group so that the RCCL
    communicator only ever carries the weight broadcast and the scalar reductions."""
    global _meta_group
    import torch.distributed as dist
    if dist.get_backend() == "gloo":
        grp = None
    else:
        if _meta_group is None:
            _meta_group = dist.new_group(backend="gloo")
        grp = _meta_group
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, row, group=grp)
    return out
