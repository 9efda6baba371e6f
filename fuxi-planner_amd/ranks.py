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


def _recv(sock, limit=None):
    def exact(n):
        buf = bytearray()
        while len(buf) < n:
            chunk = sock.recv(min(1 << 20, n - len(buf)))
            if not chunk:
                raise ConnectionError("rendezvous peer closed the connection")
            buf += chunk
        return bytes(buf)
    (n,) = struct.unpack("<Q", exact(8))
    if limit is not None and n > limit:
        raise ValueError("rendezvous message of %d bytes where at most %d were expected" % (n, limit))
    return pickle.loads(exact(n))


_HELLO, _WELCOME = "fxjps-rendezvous-hello", "fxjps-rendezvous-welcome"
_PORT_STEPS = (0, 6, 12, 100)  # the port asked for, then these offsets from it: somebody else may be listening there


class Rendezvous(object):
    """A star of TCP connections to rank 0: broadcast / gather of small Python objects, barrier, maximum.

    Rank 0 listens on `port`; if that port is taken, on the next of a short fixed list of offsets from it.  The other
    ranks try the same list in the same order and know rank 0 by its answer to their greeting, so a foreign listener on
    one of the ports (it does not answer, or answers something else) is passed over."""

    def __init__(self, rank, world, addr="127.0.0.1", port=29600, timeout=120.0):
        self.rank, self.world = int(rank), int(world)
        self.peers = []   # rank 0: sockets to ranks 1 .. world - 1, in rank order
        self.sock = None  # other ranks: the socket to rank 0
        self.port = None  # the port the star was built on
        if self.world == 1:
            return
        ports = [int(port) + d for d in _PORT_STEPS]
        if self.rank == 0:
            srv, err = None, None
            for pt in ports:
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    srv.bind((addr, pt))
                    self.port = pt
                    break
                except OSError as e:
                    err = e
                    srv.close()
                    srv = None
            if srv is None:
                raise err
            srv.listen(self.world)
            srv.settimeout(timeout)
            got = {}
            while len(got) < self.world - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(10.0)
                try:
                    hello = _recv(c, limit=4096)
                    if not (isinstance(hello, tuple) and len(hello) == 3 and hello[0] == _HELLO and int(hello[2]) == self.world
                            and 1 <= int(hello[1]) < self.world and int(hello[1]) not in got):
                        raise ValueError("not a rank of this run")
                    _send(c, (_WELCOME, self.world))
                except (OSError, ValueError, TypeError, pickle.UnpicklingError, EOFError, struct.error):
                    c.close()  # (somebody else's connection)
                    continue
                c.settimeout(None)  # (an accepted socket is blocking whatever the listener's timeout is; say so)
                got[int(hello[1])] = c
            srv.close()
            self.peers = [got[r] for r in range(1, self.world)]
        else:
            t0 = time.time()
            s = None
            while s is None:
                for pt in ports:
                    try:
                        c = socket.create_connection((addr, pt), timeout=2.0)
                    except OSError:
                        continue
                    try:
                        c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        c.settimeout(5.0)
                        _send(c, (_HELLO, self.rank, self.world))
                        ans = _recv(c, limit=4096)
                        if isinstance(ans, tuple) and len(ans) == 2 and ans[0] == _WELCOME and int(ans[1]) == self.world:
                            s, self.port = c, pt
                            break
                        c.close()
                    except (OSError, ValueError, TypeError, pickle.UnpicklingError, EOFError, struct.error):
                        c.close()  # (not rank 0: a foreign listener, or rank 0 of another run)
                if s is None:
                    if time.time() - t0 > timeout:
                        raise TimeoutError("rendezvous: rank 0 did not answer on %s ports %s within %.0f s" % (addr, ports, timeout))
                    time.sleep(0.05)
            s.settimeout(None)
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
