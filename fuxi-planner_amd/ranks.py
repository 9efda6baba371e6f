"""One process per GPU without a framework: rendezvous over a TCP socket, the grid over RCCL inside the library.

`north_star`: "no PyTorch".  The reference shares nothing between calls (jps1.py:183-192), so the ranks of a sharded run
need exactly two things from each other: the 128-byte id of the RCCL communicator (rank 0 makes it, `ncclGetUniqueId`)
and, for a benchmark, a barrier and a maximum over their timings.  Both fit a star of TCP connections to rank 0 at
MASTER_ADDR : MASTER_PORT + 1 -- the address `torch.distributed.run` (or any launcher that sets RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT) already hands every process.  The grid itself travels ONCE, device to device, by the
library's `ncclBroadcast` over xGMI (`fxjps_set_grid_rank`); results stay on their rank unless the caller gathers them.

    rdv = Rendezvous.from_env()                       # RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT (+ 1)
    rp = RankPlanner(rdv, device=local_rank)          # collective: ncclCommInitRank
    rp.set_grid(occ if rdv.rank == 0 else None)       # collective: shape over the socket, bytes over RCCL
    lo, hi, off, cells, cost, status = rp.plan_local(starts, goals)   # this rank's contiguous shard
    merged = rp.gather(off, cells, cost, status)      # rank 0: the batch in query order, byte-identical to one GPU

No torch import anywhere on this path.  `fuxi_planner_amd.distributed.ShardedPlanner` is the same thing inside an
existing `torch.distributed` process group, for callers that have one.
"""
import os
import pickle
import socket
import struct
import time

import numpy as np

from .distributed import merge_csr, shard_bounds


def _send(sock, obj):
    b = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    sock.sendall(struct.pack("<Q", len(b)) + b)


def _recv(sock):
    def exact(n):
        buf = bytearray()
        while len(buf) < n:
            chunk = sock.recv(min(1 << 20, n - len(buf)))
            if not chunk:
                raise ConnectionError("rendezvous peer closed the connection")
            buf += chunk
        return bytes(buf)
    (n,) = struct.unpack("<Q", exact(8))
    return pickle.loads(exact(n))


class Rendezvous(object):
    """A star of TCP connections to rank 0: broadcast / gather of small Python objects, barrier, maximum."""

    def __init__(self, rank, world, addr="127.0.0.1", port=29600, timeout=120.0):
        self.rank, self.world = int(rank), int(world)
        self.peers = []   # rank 0: sockets to ranks 1 .. world - 1, in rank order
        self.sock = None  # other ranks: the socket to rank 0
        if self.world == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, int(port)))
            srv.listen(self.world)
            srv.settimeout(timeout)
            got = {}
            while len(got) < self.world - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(None)  # (an accepted socket is blocking whatever the listener's timeout is; say so)
                r = _recv(c)
                got[int(r)] = c
            srv.close()
            self.peers = [got[r] for r in range(1, self.world)]
        else:
            t0 = time.time()
            while True:
                try:
                    s = socket.create_connection((addr, int(port)), timeout=5.0)
                    break
                except OSError:
                    if time.time() - t0 > timeout:
                        raise
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(None)
            _send(s, self.rank)
            self.sock = s

    @classmethod
    def from_env(cls, port_offset=1):
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
                   os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")) + port_offset)

    def bcast(self, obj=None):
        """rank 0's object on every rank"""
        if self.world == 1:
            return obj
        if self.rank == 0:
            for p in self.peers:
                _send(p, obj)
            return obj
        return _recv(self.sock)

    def gather(self, obj):
        """rank 0: the list of every rank's object in rank order; the others: None"""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            return [obj] + [_recv(p) for p in self.peers]
        _send(self.sock, obj)
        return None

    def barrier(self):
        self.bcast(self.gather(None) is not None)

    def max(self, values):
        """element-wise maximum over the ranks of a short list of floats, on every rank"""
        parts = self.gather([float(v) for v in values])
        out = [max(col) for col in zip(*parts)] if parts is not None else None
        return self.bcast(out)

    def close(self):
        for p in self.peers:
            p.close()
        if self.sock is not None:
            self.sock.close()
        self.peers, self.sock = [], None


class RankPlanner(object):
    """This rank's planner inside a `Rendezvous`.  engine_factory(device, rank, world, unique_id) -> engine lets the CPU
    test-suite drive the shard / merge logic with a checker engine; the default is the HIP planner (`Planner.for_rank`)."""

    def __init__(self, rdv, device=0, engine_factory=None, host_broadcast=False):
        self.rdv = rdv
        self.rank, self.world = rdv.rank, rdv.world
        # host_broadcast: the grid bytes go over the rendezvous socket instead of RCCL -- for ranks that share ONE device
        # (a rehearsal on a one-GPU box: RCCL refuses two ranks on a device) and for engines without a device
        self.host_broadcast = bool(host_broadcast) or engine_factory is not None
        uid = None
        if self.world > 1 and not self.host_broadcast:
            from .planner import Planner
            uid = rdv.bcast(Planner.rank_unique_id() if self.rank == 0 else None)
        self.rccl_error = None
        if engine_factory is not None:
            self.engine = engine_factory(device, self.rank, self.world, uid)
        elif self.host_broadcast:
            from .planner import Planner
            self.engine = Planner([device])
        else:
            from .planner import Planner
            from ._lib import FxjpsError
            err = None
            try:
                self.engine = Planner.for_rank(device, self.rank, self.world, uid)  # collective: ncclCommInitRank
            except FxjpsError as e:
                self.engine, err = None, str(e)
            # every rank or none: a communicator that did not come up on one rank is of no use to the others -- the grid
            # then travels over the rendezvous socket (speed of one 1 MiB message, never correctness), and the caller can
            # see why (rccl_error)
            errs = rdv.bcast(rdv.gather(err))
            if any(e is not None for e in errs):
                self.rccl_error = next(e for e in errs if e is not None)
                if self.engine is not None:
                    self.engine.close()
                self.engine = Planner([device])
                self.host_broadcast = True
        self.shape = None

    def set_grid(self, occ=None):
        """Rank 0 passes the uint8 [W][H] occupancy; every rank ends up with it resident."""
        if self.rank == 0:
            occ = np.ascontiguousarray(occ, dtype=np.uint8)
        W, H = self.rdv.bcast(tuple(occ.shape) if self.rank == 0 else None)
        if self.host_broadcast:
            data = self.rdv.bcast(occ.tobytes() if self.rank == 0 else None)
            self.engine.set_grid_occ(np.frombuffer(data, dtype=np.uint8).reshape(W, H))
        else:
            self.engine.set_grid_rank(occ if self.rank == 0 else None, W, H)  # the one collective of the path: ncclBroadcast
        self.shape = (W, H)
        return W, H

    def plan_local(self, starts, goals, hchoice=2, max_path_len=None):
        """Plan this rank's contiguous shard of the global query arrays."""
        starts = np.asarray(starts, dtype=np.int32).reshape(-1, 2)
        goals = np.asarray(goals, dtype=np.int32).reshape(-1, 2)
        lo, hi = shard_bounds(len(starts), self.rank, self.world)
        return (lo, hi) + tuple(self.engine.plan_batch(starts[lo:hi], goals[lo:hi], hchoice, max_path_len))

    def gather(self, off, cells, cost, status):
        """rank 0: the merged CSR result in query order; the others: None"""
        parts = self.rdv.gather((off, cells, cost, status))
        return merge_csr(parts) if parts is not None else None

    def close(self):
        if hasattr(self.engine, "close"):
            self.engine.close()
