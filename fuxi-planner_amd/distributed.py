"""Shard / merge arithmetic of the multi-GPU path (no framework, no device: plain numpy).

The path shards trivially: queries are independent, the grid is
read-only during a batch.  Rank r of N plans the contiguous slice
[floor(r*Q/N), floor((r+1)*Q/N)) of the query arrays, so concatenating the rank
results in rank order is byte-identical to a single-GPU run.  The only
collective on the path is one broadcast of the W*H occupancy bytes from rank 0
(RCCL over xGMI inside the library: fxjps_create / fxjps_create_rank); who calls
it is `fuxi_planner_amd.ranks` (one process per GPU, no framework in the process)
or the library's own multi-device handle.  (A wrapper for hosts that already run
a `torch.distributed` process group lives outside the package: tools/torch_group.py.)
"""
import numpy as np


def shard_bounds(nq, rank, world):
    """Contiguous shard [lo, hi) of nq queries for `rank` of `world`."""
    if not (0 <= rank < world):
        raise ValueError("rank %d not in [0, %d)" % (rank, world))
    return (nq * rank) // world, (nq * (rank + 1)) // world


def merge_csr(parts):
    """Concatenate per-rank (offsets, cells, cost, status) in rank order."""
    offs, cells, cost, status = [np.zeros(1, dtype=np.int64)], [], [], []
    base = 0
    for o, c, co, st in parts:
        offs.append(np.asarray(o[1:], dtype=np.int64) + base)
        base += int(o[-1])
        cells.append(np.asarray(c, dtype=np.int32).reshape(-1, 2))
        cost.append(np.asarray(co, dtype=np.float64))
        status.append(np.asarray(st, dtype=np.int32))
    return (np.concatenate(offs), np.concatenate(cells) if cells else np.zeros((0, 2), np.int32),
            np.concatenate(cost) if cost else np.zeros(0), np.concatenate(status) if status else np.zeros(0, np.int32))
