"""Multi-GPU host logic: one process per GPU, queries sharded, grid broadcast once.

The path shards trivially: queries are independent, the grid is
read-only during a batch.  Rank r of N plans the contiguous slice
[floor(r*Q/N), floor((r+1)*Q/N)) of the query arrays, so concatenating the rank
results in rank order is byte-identical to a single-GPU run.  The only
collective on the path is one broadcast of the W*H occupancy bytes from rank 0
(RCCL over xGMI when the process group is "nccl"); results are gathered to rank
0 only if the caller asks for it.

`torch.distributed` is plumbing here (rendezvous + the broadcast); the planner
itself never sees a torch type.  The CPU test-suite drives this module with the
gloo backend and a checker engine.
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


class ShardedPlanner(object):
    """Wraps one local engine (a `Planner` on this rank's GPU) inside an initialised
    torch.distributed process group."""

    def __init__(self, engine, device=None, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed process group is not initialised")
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.engine = engine
        self.device = device  # torch device holding the broadcast buffer ("cuda:N" with nccl, "cpu" with gloo)
        self.shape = None

    def set_grid(self, occ=None):
        """Rank 0 passes the uint8 [W][H] occupancy; every rank ends up with it resident."""
        import torch
        dist = self.dist
        dev = self.device or "cpu"
        hdr = torch.zeros(2, dtype=torch.int64, device=dev)
        if self.rank == 0:
            occ = np.ascontiguousarray(occ, dtype=np.uint8)
            hdr[0], hdr[1] = occ.shape
        dist.broadcast(hdr, 0, group=self.group)
        W, H = int(hdr[0]), int(hdr[1])
        if self.rank == 0:
            buf = torch.from_numpy(occ.reshape(-1)).to(dev)
        else:
            buf = torch.empty(W * H, dtype=torch.uint8, device=dev)
        dist.broadcast(buf, 0, group=self.group)  # the one collective of the path
        if buf.is_cuda:
            torch.cuda.synchronize(buf.device)
            self.engine.set_grid_device(buf.data_ptr(), W, H)  # the engine copies; buf may die afterwards
        else:
            self.engine.set_grid_occ(buf.numpy().reshape(W, H))
        self.shape = (W, H)
        return W, H

    def plan_local(self, starts, goals, hchoice=2, max_path_len=None):
        """Plan this rank's contiguous shard of the global query arrays."""
        starts = np.asarray(starts, dtype=np.int32).reshape(-1, 2)
        goals = np.asarray(goals, dtype=np.int32).reshape(-1, 2)
        lo, hi = shard_bounds(len(starts), self.rank, self.world)
        return (lo, hi) + tuple(self.engine.plan_batch(starts[lo:hi], goals[lo:hi], hchoice, max_path_len))

    def plan(self, starts, goals, hchoice=2, max_path_len=None):
        """Plan the whole batch; rank 0 gets the merged CSR result (others get None)."""
        lo, hi, off, cells, cost, status = self.plan_local(starts, goals, hchoice, max_path_len)
        parts = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object((off, cells, cost, status), parts, dst=0, group=self.group)
        if self.rank != 0:
            return None
        return merge_csr(parts)
