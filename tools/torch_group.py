"""A `torch.distributed` process group around the planner -- for hosts that already run one.  NOT part of the package
(`north_star`: no PyTorch in the product); the package's own multi-GPU paths are `fuxi_planner_amd.ranks` and the library's
multi-device handle.  Kept for the gloo CPU test of the shard / merge logic (tests/test_host_cpu.py) and the two-process test
around the real planner (tests/test_gpu_fullsize.py).

`torch.distributed` is plumbing here (rendezvous + the broadcast); the planner itself never sees a torch type.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fuxi_planner_amd.distributed import merge_csr, shard_bounds  # noqa: E402


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
