"""Multi-GPU sharding of the forest build: trees are independent given the read-only dataset
(src/writer.rs:556-561,798: one task per root), so each rank (= one process per GPU) builds the trees
`t = rank (mod world)` from its own replica of the dataset.  There is NO collective on the data path; the
only cross-rank traffic is the benchmark's barrier / max-of-elapsed, done by the caller with
torch.distributed (RCCL on GPUs, gloo in the CPU tests)."""
from __future__ import annotations

from typing import List, Sequence


def trees_for_rank(n_trees: int, rank: int, world: int) -> List[int]:
    """Round-robin tree indices owned by `rank`."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_trees, world))


def tree_seeds(base_seed: int, tree_indices: Sequence[int]) -> List[int]:
    """One seed per tree index, independent of how trees are sharded (so 1 GPU and 8 GPUs build the
    same forest): the analogue of `StdRng::from_seed(rng.gen())` per root task (src/writer.rs:575)."""
    out = []
    for t in tree_indices:
        x = (base_seed * 0x9E3779B97F4A7C15 + t * 0xD1B54A32D192ED03 + 0x2545F4914F6CDD1D) & 0xFFFFFFFFFFFFFFFF
        x ^= x >> 32
        out.append(x)
    return out
